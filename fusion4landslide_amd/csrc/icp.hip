// icp.hip -- batched per-patch ICP (point-to-point, point-to-plane, generalized) for gfx950.
//
// Replaces, for P patch pairs in ONE launch and with zero host round trips, the loop body
//   src/coarse_to_fine_matching_base.py:3353-3367  ->  utils/o3d_tools.py:12-71 `icp_registration`
//   ->  Open3D 0.19 registration_icp(point2point | point2plane, criteria(1e-6, 1e-6, 30))
//       (and registration_generalized_icp, the third icp_type of that function: MODE 2, f4l_piecewise_gicp)
// which in the reference copies every patch GPU->CPU, builds a KD-tree, iterates on the CPU and copies
// the 4x4 back (two device crossings per patch).
//
// Mapping to CDNA4
//   * one workgroup of NW waves (1, 2 or 4) per patch pair, every iteration inside the kernel.  The host sizes the LDS
//     for the largest patch of a launch; batches of uneven sizes are binned on the device and launched per size class,
//     side by side on helper streams (icp_launch_host);
//   * the target patch is counting-sorted ONCE into an LDS-resident uniform grid (patch_grid.h: cells of the search
//     radius, finer for patches that are dense relative to it) and stays there for every iteration;
//   * a pass has two phases.  Phase 1 re-measures every source point's previous correspondence and certifies it
//     (triangle inequality against the runner-up distance of its last search and the point's own motion since) or
//     queues the point; phase 2 searches the queued points only, 64 per wave, each lane walking the <= 9 grid runs
//     its bound reaches in one flat predicated loop.  ~87 % of the point-iterations are certified;
//   * the correspondence sums of a pass (17 for Umeyama, 29 for the 6x6 point-to-plane system) are reduced by
//     row_sums_transposed (f4l_device.h: the DPP quad stages transpose while they add), row partials go through LDS,
//     one wave -- rotating with the patch index -- adds them, solves (Newton on SO(3), warm-started Jacobi SVD as
//     fallback; elimination for the 6x6) and publishes the new transform through LDS;
//   * convergence test, iteration count, fitness and rmse on the device; optional Kabsch initialisation from
//     correspondences in the prologue and displacement rows in the epilogue (f4l_patch_loop).
// The kernel is bound by the latency of that per-patch chain, not by HBM: per patch both clouds are read
// about once; the ALGORITHMIC traffic the roofline is priced with stays the reference's dataflow, 24 B per source
// point per iteration.
//
// Numerics: coordinates are taken relative to a per-patch origin (first target point); the running transform, the
// sums from the 16-lane rows on and the solves are double.  F = double (F4L_SEARCH_F64, the default) evaluates positions
// and squared distances in double on the original float32 coordinates, like Open3D: same trajectory as the oracle to
// 1e-9 m.  F = float searches in float32 on patch-relative coordinates.  Either way the search returns exactly the
// brute-force answer of its arithmetic: the minimiser of (d2, index).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include <mutex>

#include "f4l_device.h"
#include "ldlt6.h"
#include "patch_grid.h"

namespace f4l {

#ifndef ICP_WAVES_PER_EU
#define ICP_WAVES_PER_EU 4
#endif
#ifndef ICP_WIDE_WPE
#define ICP_WIDE_WPE 3  // (the headline shape: 168 registers; 2 and 4 measured in round 6, profiles/r6_an_*)
#endif
#ifndef ICP_WIDE_PLANE_WPE
#define ICP_WIDE_PLANE_WPE 3  // (the plane estimators' bulk shape; 2 = 256 registers: measured)
#endif
#ifndef ICP_WAVES_PER_EU_F64
#define ICP_WAVES_PER_EU_F64 4
#endif
constexpr int ICP_LDS_BUDGET = 160 * 1024 - 512;  // dynamic LDS a single workgroup may ask for on gfx950
constexpr int ICP_TGT_MAX = 8192;                 // target points kept in LDS at most
constexpr int ICP_CELL_MAX = 16384;               // grid cells at most (uint16 prefix table)

struct IcpArgs {
    const float *src;
    const int64_t *src_off;
    const float *tgt;
    const int64_t *tgt_off;
    int64_t P;
    const double *init_T;
    // fused loop body (f4l_patch_loop): Kabsch initialisation from correspondences in the prologue, displacement rows
    // in the epilogue; all nullable
    const float *corr_src, *corr_ref, *corr_w;
    const int64_t *corr_off;
    double kabsch_w_thresh, kabsch_eps;
    float *rows_out;
    // rows for another cloud than the one ICP runs on (the reference registers the MUTUAL points of a patch match and applies
    // the transform to ALL points of the source patch, src/coarse_to_fine_matching_base.py:3348-3374); null: the ICP cloud
    const float *rows_src;
    const int64_t *rows_off;
    int64_t min_corr;  // patch matches with fewer correspondences are skipped (:3338, `num_min_fine_match`)
    int init_round_f32;  // the Kabsch transform reaches ICP as float32 values (scripts/weighted_svd.py:148-151: a float32 4 x 4)
    const float *tgt_normals;
    int normals_f64;  // tgt_normals points at doubles (F4L_ICP_NORMALS_F64)
    int p2pl_open3d;  // point-to-plane steps with Open3D's own semantics (F4L_ICP_P2PL_OPEN3D), see p2plane_step_open3d
    // generalized ICP (F4L_ICP_GENERALIZED, f4l_piecewise_gicp): the source's normals (double [n_src][3]; the target's through
    // tgt_normals, double) and the estimator's epsilon
    const double *src_normals;
    double gicp_eps;
    double r, r2;
    int max_iter;
    double rel_fitness, rel_rmse;
    int fixed_iters;
    int tgt_cap;   // target slots in LDS (patches with more targets take the brute-force global path)
    int cert_cap;  // source points per patch the certificate arrays (prev, mabs, queue) can hold
    int src_cap;   // source points per patch staged in LDS (0: read from global memory every pass)
    int pp_cap;    // source points per patch whose position at their last search is kept (per-point certificates)
    int cell_cap;  // grid cells the prefix table can hold
    double *T_out, *fitness_out, *rmse_out;
    int32_t *iters_out, *corr_out;
    int subdiv;     // cells per radius the grid may use (patch_grid.h: grid_build)
    int xsub;       // x-cells per cell edge at most (1, 2, 4, 8: patch_grid.h "Cell shape")
    float dens;     // points per bounding-box cell a subdivided grid keeps on average
    double mu_frac; // certificate margin as a fraction of the correspondence radius ...
    double mu_cell; // ... and of the cell edge (the smaller of the two counts: fine grids, i.e. dense patches, get less)
    double mu_cell_fine;  // ... of the cell edge on grids finer than the radius (wmax > 1)
    int debug;  // F4L_ICP_DEBUG env, bit switches for A/B measurements and tests: 4 = no certificates, 8 = no bound from
                // the previous correspondence, 16 = no narrow look-up before pass 0 on fine grids, 128 = always the Jacobi
                // SVD (no Newton), 64 = search counters (profiling build), 512 = no column grids (layers of cells in z for
                // flat patches too), 1024 = no finer cells along x (512 + 1024: the cubic cells of rounds 1-3)
    unsigned long long *prof;  // F4L_ICP_PROF builds only: per-phase shader-clock totals (see f4l_piecewise_icp)
    // size-class launches (icp_launch_host): workgroup b handles patch list[b] if b < *list_cnt, else nothing
    const int *list, *list_cnt;
    int prof_max_n;  // profiling builds: only patches up to this size are counted
};

#ifdef F4L_ICP_PROF
#define PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD(slot, t1, t0) do { prof_acc[slot] += (unsigned long long)((t1) - (t0)); } while (0)
#define PROF_CNT(slot, v) do { if (a.prof) atomicAdd(&a.prof[slot], (unsigned long long)(v)); } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, t1, t0)
#define PROF_CNT(slot, v)
#endif

// Brute-force fallback with targets in global memory (patches larger than the LDS budget): wave-uniform
// addresses, ascending index, so the (d2, index) minimiser is the same as the grid's.
template <typename F>
__device__ __forceinline__ void nn_global(const float *__restrict__ tg, int nt, float ox, float oy, float oz, F px, F py,
                                          F pz, Best<F> &best) {
#pragma unroll 4
    for (int j = 0; j < nt; ++j) {
        const F qx = (F)tg[3 * j] - (F)ox, qy = (F)tg[3 * j + 1] - (F)oy, qz = (F)tg[3 * j + 2] - (F)oz;
        best.offer(grid_d2(px - qx, py - qy, pz - qz), (unsigned int)j);  // tag = plain index (no slot on this path)
    }
}

// Solve the 6x6 system M[:, :6] x = M[:, 6] in place by Gaussian elimination with partial pivoting.
// All indices are compile-time after unrolling, so M lives in registers.  Returns false when singular -- a pivot that is
// zero, or below 1e-13 of the largest diagonal entry: the system is built about the patch's own origin, where a patch that
// pins its six degrees of freedom keeps its pivots above 1e-6 of that; below the bound the "solution" is rounding noise
// times 1e13 (a patch left with four correspondences was thrown 55 m by it), and the step is not taken.
__device__ __forceinline__ bool solve6(double (&M)[6][7], double (&x)[6]) {
    bool ok = true;
    double dmax = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) dmax = fmax(dmax, fabs(M[c][c]));
    const double tiny = 1e-13 * dmax;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            if (fabs(M[r][c]) > fabs(M[c][c])) {
#pragma unroll
                for (int j = 0; j < 7; ++j) { const double t = M[c][j]; M[c][j] = M[r][j]; M[r][j] = t; }
            }
        }
        const double piv = M[c][c];
        if (!(fabs(piv) > tiny) || !isfinite(piv)) ok = false;
        const double ip = 1.0 / piv;
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] * ip;
#pragma unroll
            for (int j = c; j < 7; ++j) M[r][j] -= f * M[c][j];
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double s = M[r][6];
#pragma unroll
        for (int j = r + 1; j < 6; ++j) s -= M[r][j] * x[j];
        x[r] = s / M[r][r];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) if (!isfinite(x[i])) ok = false;
    return ok;
}

// The point-to-plane system is symmetric positive definite (J^T J): L D L^T WITHOUT pivoting (Open3D's own solve is Eigen's
// ldlt, which pivots on the diagonal: ldlt6.h, the F4L_ICP_P2PL_OPEN3D step; on the well-conditioned systems this one accepts
// the pivot order does not show) -- a quarter of the instructions of the pivoted elimination above, which the solving wave
// issues once per patch and pass while the rest of the workgroup waits.  M[:, :6] is read in its lower triangle, M[:, 6] is the right-hand side.  Returns
// false when a pivot is not above 1e-13 of the largest diagonal entry (see solve6) or anything is not finite.
__device__ __forceinline__ bool solve6_spd(const double (&M)[6][7], double (&x)[6]) {
    double dmax = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) dmax = fmax(dmax, fabs(M[c][c]));
    const double tiny = 1e-13 * dmax;
    bool ok = true;
    double L[6][6], W[6][6], D[6];  // W[i][k] = L[i][k] D[k]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * W[j][k];
        D[j] = dj;
        if (!(dj > tiny) || !isfinite(dj)) ok = false;
        const double inv = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * W[j][k];
            W[i][j] = v;
            L[i][j] = v * inv;
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = M[i][6];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i] / D[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
        x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) if (!isfinite(x[i])) ok = false;
    return ok;
}

// sin and cos of a step's Euler angle: below half a radian (every step but a wild one) the Taylor series to x^15 / x^16, whose
// next terms are below 1e-19; the library pair, with its range reduction, costs five times the instructions.
__device__ __forceinline__ void sincos_step(double x, double &sn, double &cs) {
    if (fabs(x) < 0.5) {
        const double x2 = x * x;
        double p = -1.0 / 1307674368000.0;
        p = p * x2 + 1.0 / 6227020800.0; p = p * x2 - 1.0 / 39916800.0; p = p * x2 + 1.0 / 362880.0;
        p = p * x2 - 1.0 / 5040.0; p = p * x2 + 1.0 / 120.0; p = p * x2 - 1.0 / 6.0;
        sn = x + x * (x2 * p);
        double q = 1.0 / 20922789888000.0;
        q = q * x2 - 1.0 / 87178291200.0; q = q * x2 + 1.0 / 479001600.0; q = q * x2 - 1.0 / 3628800.0;
        q = q * x2 + 1.0 / 40320.0; q = q * x2 - 1.0 / 720.0; q = q * x2 + 1.0 / 24.0;
        cs = 1.0 - x2 * (0.5 - x2 * q);
    } else { sn = sin(x); cs = cos(x); }
}

// The point-to-plane step with Open3D's own semantics (F4L_ICP_P2PL_OPEN3D; utils/o3d_tools.py:38-39,46-50 ->
// TransformationEstimationPointToPlane::ComputeTransformation -> SolveLinearSystemPSD, checks off): the system, summed about
// the patch origin for accuracy, is moved to the CALLER's origin -- the frame Open3D builds it in, which decides what the
// pseudo-inverse does to a singular system -- and solved by Eigen's diagonal-pivoted L D L^T (ldlt6.h), whatever its rank.
// Reads the row partials of the pass from LDS (rows x 29 doubles at byte offset scratch_off, added in the order the kernel adds
// them) and leaves x[0..6) at state[40..46).  Not inlined: the default path's register allocation must not see it.
__device__ __noinline__ void p2plane_step_open3d(int scratch_off, int rows, int state_off, float ox, float oy, float oz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const double *scratch = reinterpret_cast<const double *>(smem_raw + scratch_off);
    double *state = reinterpret_cast<double *>(smem_raw + state_off);
    double M[6][6], b[6], x[6];
    auto total = [&](int v) {
        double t = scratch[v];
        for (int w = 1; w < rows; ++w) t += scratch[w * 29 + v];
        return t;
    };
    int k = 2;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = u; v < 6; ++v) { const double t = total(k++); M[u][v] = t; M[v][u] = t; }
#pragma unroll
    for (int u = 0; u < 6; ++u) b[u] = -total(23 + u);
    p2plane_system_to_caller_frame(M, b, (double)ox, (double)oy, (double)oz);
    ldlt6_solve_eigen(M, b, x);
    if (lane_id() == 0) {
#pragma unroll
        for (int u = 0; u < 6; ++u) state[40 + u] = x[u];
    }
}

// uniform double held in LDS -> scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// LDS layout (dynamic):
//   scratch : NW * 32 doubles (partial sums; doubles as the scratch of grid_build and of the prologue reductions)
//   state   : 48 doubles: Rc[9] 0..8, tc[3] 9..11, done 12, fitness 13, rmse 14, iterations 15, V[9] of the last
//             SVD 16..24 (warm start), source centroid 25..27, source radius 28, accumulated motion bound 29
//   qcnt    : NW ints (length of each wave's segment of the search queue)
//   tl      : tgt_cap grid points          mabs : cert_cap F        rl : (GRID_ROWS + 1) * NT uint
//   E       : cell_cap + 8 uint16          prev : cert_cap uint16   queue : NW * seg uint16
//
// Nearest-neighbour certificates.  A search that scans everything within sqrt(b0) of the query knows, besides
// the nearest target, a distance M that every OTHER target keeps: M^2 = min(runner-up d2, b0).  While the query
// has moved less than M - d(query, nearest) since then, the nearest target is provably unchanged (triangle
// inequality) and the search is skipped: only d(query, nearest) is re-measured.  Motion is bounded per pass for
// the whole patch by |p_new - p_old| <= ||Ru - I||_F * radius + |(Ru - I) centroid + tu| and summed in state[29];
// mabs[i] stores M + (that sum at search time).  Each pass therefore has two phases: a dense one that re-measures
// and certifies (or queues) every source point, and a search phase over the queued points only, compacted
// across the workgroup.  Results are exactly those of searching every point in every pass.
// Grids finer than the radius (PatchGrid::wmax > 1): before the first pass, look for every point's neighbour inside the
// 3 x 3 rows of its own cell only (bound just under one cell edge).  Points that find one enter pass 0 with a
// correspondence and a certificate like in any later pass; only the others pay for the full (2 wmax + 1)^2 stencil,
// and they do it in dense batches of their own (phase 2).  Runs once per patch, outside the pass loop, and is
// deliberately not inlined: the loop's register allocation must not see it.
template <typename F, int NT> struct PrepassArgs {
    PatchGrid<F> g;
    // byte offsets into the workgroup's dynamic LDS (pointers rebuilt inside, so that the accesses stay ds_* ops)
    int tl, E, rl, sl /* -1: sources are read from global memory */, state, prev, mabs, ps /* -1: patch-wide bound */;
    const float *sg;  // source points in global memory
    float ox, oy, oz;
    int ns, nt;
    F rs;
};

template <typename F, int NT>
__device__ __noinline__ void icp_prepass(const PrepassArgs<F, NT> &q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = (int)threadIdx.x;
    const GridPt<F> *tl = reinterpret_cast<const GridPt<F> *>(smem_raw + q.tl);
    const unsigned short *E = reinterpret_cast<const unsigned short *>(smem_raw + q.E);
    unsigned int *rl = reinterpret_cast<unsigned int *>(smem_raw + q.rl);
    const F *sl = reinterpret_cast<const F *>(smem_raw + (q.sl < 0 ? 0 : q.sl));
    const double *state = reinterpret_cast<const double *>(smem_raw + q.state);
    unsigned short *prev = reinterpret_cast<unsigned short *>(smem_raw + q.prev);
    float *mabs = reinterpret_cast<float *>(smem_raw + q.mabs);
    float *ps = reinterpret_cast<float *>(smem_raw + (q.ps < 0 ? 0 : q.ps));
    const F R0 = (F)state[0], R1 = (F)state[1], R2 = (F)state[2], R3 = (F)state[3], R4 = (F)state[4],
            R5 = (F)state[5], R6 = (F)state[6], R7 = (F)state[7], R8 = (F)state[8];
    const F t0f = (F)state[9], t1f = (F)state[10], t2f = (F)state[11];
    for (int base = 0; base < q.ns; base += NT) {
        const int i = base + tid;
        const int ii = i < q.ns ? i : q.ns - 1;
        F x, y, z;
        if (q.sl >= 0) { x = sl[3 * ii]; y = sl[3 * ii + 1]; z = sl[3 * ii + 2]; }
        else { x = (F)q.sg[3 * ii] - (F)q.ox; y = (F)q.sg[3 * ii + 1] - (F)q.oy; z = (F)q.sg[3 * ii + 2] - (F)q.oz; }
        const F px = R0 * x + R1 * y + R2 * z + t0f;
        const F py = R3 * x + R4 * y + R5 * z + t1f;
        const F pz = R6 * x + R7 * y + R8 * z + t2f;
        F bt = grid_narrow_bound(q.g, px, py, pz);
        bt = bt < q.rs ? bt : q.rs;
        const bool valid = i < q.ns && bt > (F)0;
        const F b0 = valid ? bt * bt : (F)0;
        Best<F> best;
        best.init(b0);
        grid_nn<F, NT, false>(q.g, tl, q.nt, E, rl, valid, px, py, pz, best);
        if (valid && best.found()) {
            const F m2 = best.second < b0 ? best.second : b0;
            prev[i] = (unsigned short)best.slot();
            // (the patch-wide motion sum state[29] is still zero here)
            mabs[i] = (float)(grid_sqrt<F>(m2) * (F)(q.ps >= 0 ? 0.999999 : 0.999999 * 0.9999998));
            if (q.ps >= 0) { ps[3 * i] = (float)px; ps[3 * i + 1] = (float)py; ps[3 * i + 2] = (float)pz; }
        }
    }
    // (every thread wrote its own points only, and phase 1 reads them with the same thread: no barrier)
}

// WIDE = the register budget of three waves per SIMD (170 VGPRs) instead of four (128): the float64 build keeps ~50 VGPRs
// in scratch at 128.  Worth it only for two-wave workgroups in the throughput regime (see icp_shape).
// The sweep of a pass (phases 1 and 2 of icp_kernel) as a block, or -- in the throughput shapes' translation units -- as a generic lambda
// instantiated for both ways of summing (DEFER: see the kernel).  F4L_ICP_DEBUG bit 2048: sum where the points are found (the A/B).
#ifdef F4L_ICP_BULK_TU
#define ICP_SWEEP_OPEN auto sweep = [&](auto defer_tag) { constexpr bool DEFER = decltype(defer_tag)::value;
#define ICP_SWEEP_CLOSE                                                                   \
    };                                                                                    \
    if constexpr (MODE == F4L_ICP_POINT2PLANE) {                                          \
        if (use_cert && !(a.debug & 2048)) sweep(std::true_type{});                       \
        else sweep(std::false_type{});                                                    \
    } else {                                                                              \
        sweep(std::false_type{});                                                         \
    }
#else
#define ICP_SWEEP_OPEN { constexpr bool DEFER = false;
#define ICP_SWEEP_CLOSE }
#endif
template <int MODE, int NW, typename F, bool WIDE = false>
__global__ __launch_bounds__(NW * 64, WIDE ? (MODE == F4L_ICP_POINT2POINT ? ICP_WIDE_WPE : ICP_WIDE_PLANE_WPE) : (sizeof(F) == 8 ? ICP_WAVES_PER_EU_F64 : ICP_WAVES_PER_EU)) void icp_kernel(IcpArgs a) {
    constexpr int NV = (MODE == F4L_ICP_POINT2POINT) ? 17 : 29;
    constexpr int NT = NW * 64;
    // Correspondence sums.  float32 search + point-to-point: per-lane partial sums and the in-wave reduction are
    // float32 on coordinates CENTRED on the image of the source centroid (|values| <= patch radius, means ~ 0, so
    // neither the products nor the covariance's mean correction cancel); everything after the sums of the 16-lane
    // rows is double.  float64 search (parity mode) and point-to-plane: double throughout, uncentred like the reference.
    constexpr bool CENTRED = (MODE == F4L_ICP_POINT2POINT) && sizeof(F) == 4;
    using A = typename std::conditional<CENTRED, float, double>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // partial sums: one row of NV doubles per 16-lane row of every wave (row_sums_transposed); icp_plan() sizes the
    // region the same way
    constexpr int SUM_ROWS = 4 * NW;
    constexpr int SCRATCH = (SUM_ROWS * NV + 1) & ~1;
    double *scratch = reinterpret_cast<double *>(smem_raw);
    double *state = scratch + SCRATCH;
    int *qcnt = reinterpret_cast<int *>(state + 48);
    GridPt<F> *tl = reinterpret_cast<GridPt<F> *>(qcnt + 4);
    float *mabs = reinterpret_cast<float *>(tl + a.tgt_cap + 1);  // tl[nt] is the dummy record of the grid
    float *ps = mabs + ((a.cert_cap + 3) & ~3);  // position of every source point when it was last searched (float32)
    F *sl = reinterpret_cast<F *>(ps + 3 * ((a.pp_cap + 3) & ~3));  // origin-relative source points, packed xyz
    unsigned int *rl = reinterpret_cast<unsigned int *>(sl + 3 * ((a.src_cap + 3) & ~3));
    unsigned short *E = reinterpret_cast<unsigned short *>(rl + (GRID_ROWS + 1) * NT);
    unsigned short *prev = E + a.cell_cap + 8;
    const int seg = ((a.cert_cap + NT - 1) / NT) * 64;  // queue entries one wave can produce
    unsigned short *queue = prev + ((a.cert_cap + 7) & ~7);

    int64_t p = blockIdx.x;
    if (a.list) {
        if ((int)blockIdx.x >= *a.list_cnt) return;
        p = a.list[blockIdx.x];
    }
    if (p >= a.P) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = a.src_off[p], t0 = a.tgt_off[p];
    const int ns = (int)(a.src_off[p + 1] - s0), nt = (int)(a.tgt_off[p + 1] - t0);
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    // a patch match with too few correspondences is not registered at all (:3338-3436: `mask_spt_match_global[i] = False`)
    const bool skipped = a.corr_off != nullptr && a.corr_off[p + 1] - a.corr_off[p] < a.min_corr;
    const bool active = ns > 0 && a.r2 > 0.0 && !skipped;  // o3d returns the init untouched when max_corr_dist <= 0
    const bool tgt_in_lds = nt > 0 && nt <= a.tgt_cap;
    const bool use_cert = tgt_in_lds && ns <= a.cert_cap && !(a.debug & 4);
    const bool src_in_lds = ns <= a.src_cap;
    // per-point certificates measure each point's own displacement since its last search; without room for that they
    // fall back to the patch-wide motion bound accumulated in state[29]
    const bool per_point = use_cert && ns <= a.pp_cap;

    // per-patch origin: first target point (else first source point, else 0)
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    else if (ns > 0) { ox = sg[0]; oy = sg[1]; oz = sg[2]; }

#ifdef F4L_ICP_PROF
    unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef F4L_ICP_PROF
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
    PROF_T(pt_start);
    // search radius: a little beyond the correspondence radius, so that "no target within r" can be certified too
    const F rF = (F)a.r, r2 = (F)a.r2;
    const F rs = rF * (F)1.0625, rs2 = rs * rs;
    PatchGrid<F> g;
    g.minx = g.miny = g.minz = (F)0; g.h = (F)1; g.inv_h = (F)1; g.inv_hx = (F)1; g.inv_hz = (F)1; g.nx = g.ny = g.nz = 1; g.xs = 1; g.wmax = 1;
    double cs[3] = {0.0, 0.0, 0.0}, srad = 0.0;
    if (active) {  // uniform across the workgroup
        if (tgt_in_lds) grid_build<F, NT>(tg, nt, ox, oy, oz, rs, a.cell_cap, tl, E, reinterpret_cast<F *>(scratch), g, a.subdiv, (F)a.dens,
                                            (a.debug & 1024) ? 1 : a.xsub, !(a.debug & 512));
        {
            // centroid and radius of the source patch (origin-relative): the lever arm of the motion bound
            double sum[3] = {0.0, 0.0, 0.0};
            for (int i = tid; i < ns; i += NT) {
                sum[0] += (double)((F)sg[3 * i] - (F)ox); sum[1] += (double)((F)sg[3 * i + 1] - (F)oy);
                sum[2] += (double)((F)sg[3 * i + 2] - (F)oz);
                if (use_cert) { prev[i] = 0xffffu; mabs[i] = 0.f; }
                if (src_in_lds) { sl[3 * i] = (F)sg[3 * i] - (F)ox; sl[3 * i + 1] = (F)sg[3 * i + 1] - (F)oy; sl[3 * i + 2] = (F)sg[3 * i + 2] - (F)oz; }
            }
            block_sum<3, NW>(sum, scratch);
            cs[0] = sum[0] / (double)ns; cs[1] = sum[1] / (double)ns; cs[2] = sum[2] / (double)ns;
            double mx2 = 0.0;
            for (int i = tid; i < ns; i += NT) {
                const double dx = (double)((F)sg[3 * i] - (F)ox) - cs[0], dy = (double)((F)sg[3 * i + 1] - (F)oy) - cs[1],
                             dz = (double)((F)sg[3 * i + 2] - (F)oz) - cs[2];
                const double d2 = dx * dx + dy * dy + dz * dz;
                mx2 = d2 > mx2 ? d2 : mx2;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(mx2, m, 64); mx2 = o > mx2 ? o : mx2; }
            if (NW > 1) {
                __syncthreads();
                if (lane == 0) scratch[wave] = mx2;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) mx2 = scratch[w] > mx2 ? scratch[w] : mx2;
            }
            srad = sqrt(mx2) * (1.0 + 1e-6);
        }
    }
    __syncthreads();  // scratch is free again
    // margin of the certificates beyond the re-measured correspondence: a fraction of the radius, less on grids
    // finer than the radius (dense patches), where a wide margin would pull many points into every search
    const F mcell = g.h * (F)(g.wmax > 1 ? a.mu_cell_fine : a.mu_cell);
    const F mu = rF * (F)a.mu_frac < mcell ? rF * (F)a.mu_frac : mcell;

    // Fused initialisation: weighted Kabsch of this patch's correspondences (scripts/weighted_svd.py:58-129, the same
    // arithmetic as kabsch_kernel<float, NW, false>): two streaming passes, block reductions, SVD on one thread.
    double Tk[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // [R | t] rows
    const bool fused_init = a.corr_off != nullptr;
    if (fused_init) {
        const int64_t c0 = a.corr_off[p];
        const int nc = skipped ? 0 : (int)(a.corr_off[p + 1] - c0);
        const float *__restrict__ ks = a.corr_src + 3 * c0, *__restrict__ kr = a.corr_ref + 3 * c0;
        const float *__restrict__ kw = a.corr_w ? a.corr_w + c0 : nullptr;
        double s7[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < nc; i += NT) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            s7[0] += wi;
            s7[1] += wi * (double)ks[3 * i]; s7[2] += wi * (double)ks[3 * i + 1]; s7[3] += wi * (double)ks[3 * i + 2];
            s7[4] += wi * (double)kr[3 * i]; s7[5] += wi * (double)kr[3 * i + 1]; s7[6] += wi * (double)kr[3 * i + 2];
        }
        block_sum<7, NW>(s7, scratch);
        const double inv = 1.0 / (s7[0] + a.kabsch_eps);  // eps stays in the denominator (weighted_svd.py:96)
        const double k0 = s7[1] * inv, k1 = s7[2] * inv, k2 = s7[3] * inv;
        const double l0 = s7[4] * inv, l1 = s7[5] * inv, l2 = s7[6] * inv;
        double h9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < nc; i += NT) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            wi *= inv;
            const double a0 = (double)ks[3 * i] - k0, a1 = (double)ks[3 * i + 1] - k1, a2 = (double)ks[3 * i + 2] - k2;
            const double b0 = wi * ((double)kr[3 * i] - l0), b1 = wi * ((double)kr[3 * i + 1] - l1),
                         b2 = wi * ((double)kr[3 * i + 2] - l2);
            h9[0] += a0 * b0; h9[1] += a0 * b1; h9[2] += a0 * b2;
            h9[3] += a1 * b0; h9[4] += a1 * b1; h9[5] += a1 * b2;
            h9[6] += a2 * b0; h9[7] += a2 * b1; h9[8] += a2 * b2;
        }
        __syncthreads();
        block_sum<9, NW>(h9, scratch);
        if (tid == 0 && nc > 0) {
            double R[9];
            const double Ht[9] = {h9[0], h9[3], h9[6], h9[1], h9[4], h9[7], h9[2], h9[5], h9[8]};
            if (!rot_newton(Ht, R)) {  // R = V diag(1,1,sign det(V U^T)) U^T of H = U S V^T (weighted_svd.py:107-111)
                double U[9], V[9];
                const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                svd3_warm(h9, I3, U, V);
                const double dd = det3(V) * det3(U);
                mul_diag_bt(V, dd > 0.0 ? 1.0 : (dd < 0.0 ? -1.0 : 0.0), U, R);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { Tk[4 * i] = R[3 * i]; Tk[4 * i + 1] = R[3 * i + 1]; Tk[4 * i + 2] = R[3 * i + 2]; }
            Tk[3] = l0 - (R[0] * k0 + R[1] * k1 + R[2] * k2);
            Tk[7] = l1 - (R[3] * k0 + R[4] * k1 + R[5] * k2);
            Tk[11] = l2 - (R[6] * k0 + R[7] * k1 + R[8] * k2);
            if (a.init_round_f32) {
#pragma unroll
                for (int i = 0; i < 12; ++i) Tk[i] = (double)(float)Tk[i];
            }
        }
        __syncthreads();
    }

    // running transform in origin-relative coordinates, p' = Rc s' + tc, lives in LDS `state`
    if (tid == 0) {
        if (fused_init || a.init_T) {
            const double *T = fused_init ? Tk : a.init_T + 16 * p;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                state[3 * i] = T[4 * i]; state[3 * i + 1] = T[4 * i + 1]; state[3 * i + 2] = T[4 * i + 2];
                state[9 + i] = T[4 * i] * (double)ox + T[4 * i + 1] * (double)oy + T[4 * i + 2] * (double)oz +
                               T[4 * i + 3] - (double)(i == 0 ? ox : (i == 1 ? oy : oz));
            }
        } else {
            state[0] = 1; state[1] = 0; state[2] = 0; state[3] = 0; state[4] = 1; state[5] = 0;
            state[6] = 0; state[7] = 0; state[8] = 1; state[9] = 0; state[10] = 0; state[11] = 0;
        }
        state[12] = 0.0; state[13] = 0.0; state[14] = 0.0; state[15] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) state[16 + i] = (i % 4 == 0) ? 1.0 : 0.0;
        state[25] = cs[0]; state[26] = cs[1]; state[27] = cs[2]; state[28] = srad; state[29] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)  // image of the source centroid under the current transform
            state[30 + i] = state[3 * i] * cs[0] + state[3 * i + 1] * cs[1] + state[3 * i + 2] * cs[2] + state[9 + i];
    }
    __syncthreads();
    PROF_T(pt_built);
    PROF_ADD(1, pt_built, pt_start);

    if (active && use_cert && g.wmax > 1 && !(a.debug & 16)) {
        PrepassArgs<F, NT> q;
        auto lds_off = [&](const void *ptr) { return (int)((const unsigned char *)ptr - smem_raw); };
        q.g = g; q.tl = lds_off(tl); q.E = lds_off(E); q.rl = lds_off(rl); q.sl = src_in_lds ? lds_off(sl) : -1;
        q.state = lds_off(state); q.prev = lds_off(prev); q.mabs = lds_off(mabs); q.ps = per_point ? lds_off(ps) : -1;
        q.sg = sg; q.ox = ox; q.oy = oy; q.oz = oz; q.ns = ns; q.nt = nt; q.rs = rs;
        icp_prepass<F, NT>(q);
    }

    const int n_pass = active ? a.max_iter + 1 : 0;
    // the solving wave rotates with the patch index so that co-resident workgroups do not all solve on the same SIMD
    const int solver = (int)(blockIdx.x % NW);

    float nfx = 0.f, nfy = 0.f, nfz = 0.f;  // phase 1's next batch of source points (see there)
    if (active && use_cert && !src_in_lds) {
        const int in = tid < ns ? tid : ns - 1;
        nfx = sg[3 * in]; nfy = sg[3 * in + 1]; nfz = sg[3 * in + 2];
    }
    for (int pass = 0; pass < n_pass; ++pass) {
        // the transform is uniform: keep it in scalar registers
        const F R0 = (F)uniform_f64(state[0]), R1 = (F)uniform_f64(state[1]), R2 = (F)uniform_f64(state[2]),
                R3 = (F)uniform_f64(state[3]), R4 = (F)uniform_f64(state[4]), R5 = (F)uniform_f64(state[5]),
                R6 = (F)uniform_f64(state[6]), R7 = (F)uniform_f64(state[7]), R8 = (F)uniform_f64(state[8]);
        const F t0f = (F)uniform_f64(state[9]), t1f = (F)uniform_f64(state[10]), t2f = (F)uniform_f64(state[11]);
        const F dsum = (F)(uniform_f64(state[29]) * (1.0 + 1e-6));  // rounded up
        const F cpx = CENTRED ? (F)uniform_f64(state[30]) : (F)0, cpy = CENTRED ? (F)uniform_f64(state[31]) : (F)0,
                cpz = CENTRED ? (F)uniform_f64(state[32]) : (F)0;
        A acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = (A)0;
        PROF_T(pt_p0);

        // one accepted correspondence (p: moved source point, q: target, d: squared distance, bj: target index)
        // (si: the source point's index inside its patch -- generalized ICP reads its normal)
        auto accumulate = [&](A (&acc)[NV], F px, F py, F pz, F qx, F qy, F qz, F d, int bj, int si) {
            (void)si;
            const A dpx = (A)(px - cpx), dpy = (A)(py - cpy), dpz = (A)(pz - cpz);
            const A dqx = (A)(qx - cpx), dqy = (A)(qy - cpy), dqz = (A)(qz - cpz);
            acc[0] += (A)1;
            acc[1] += (A)d;
            if (MODE == F4L_ICP_POINT2POINT) {
                acc[2] += dpx; acc[3] += dpy; acc[4] += dpz;
                acc[5] += dqx; acc[6] += dqy; acc[7] += dqz;
                acc[8] += dqx * dpx; acc[9] += dqx * dpy; acc[10] += dqx * dpz;
                acc[11] += dqy * dpx; acc[12] += dqy * dpy; acc[13] += dqy * dpz;
                acc[14] += dqz * dpx; acc[15] += dqz * dpy; acc[16] += dqz * dpz;
            } else if (MODE == F4L_ICP_GENERALIZED) {
                // Open3D's TransformationEstimationForGeneralizedICP (GeneralizedICP.cpp): per pair M = C_q + C_s, three residual
                // rows W (p - q) with W = M^-1/2 and Jacobian rows W [-[p]x | I].  Only W^T W = M^-1 enters the normal
                // equations: J^T J = A^T M^-1 A, J^T r = A^T M^-1 d with A = [-[p]x | I] -- no matrix square root here (the
                // oracle takes it, as Open3D does).  The covariances come from the normals (InitializePointCloudFor
                // GeneralizedICP: Rx diag(eps, 1, 1) Rx^T = I - (1 - eps) n n^T for a unit normal; GetRotationFromE1ToX
                // returns the identity for n.x < -0.99, i.e. such a point gets e1's covariance), and the source's turns with
                // the cloud (PointCloud::Transform): R C_s R^T = I - (1 - eps) (R n)(R n)^T with the accumulated rotation.
                // Rows p x w and w, like point-to-plane's: the system moves to the caller's origin the same way.
                const double *nq = reinterpret_cast<const double *>(a.tgt_normals) + 3 * (t0 + bj);
                const double *np_ = a.src_normals + 3 * (s0 + si);
                double tx = nq[0], ty = nq[1], tz = nq[2];
                if (tx < -0.99) { tx = 1.0; ty = 0.0; tz = 0.0; }
                double sx = np_[0], sy = np_[1], sz = np_[2];
                if (sx < -0.99) { sx = 1.0; sy = 0.0; sz = 0.0; }
                const double ux = (double)R0 * sx + (double)R1 * sy + (double)R2 * sz;
                const double uy = (double)R3 * sx + (double)R4 * sy + (double)R5 * sz;
                const double uz = (double)R6 * sx + (double)R7 * sy + (double)R8 * sz;
                const double k = 1.0 - a.gicp_eps;
                const double m00 = 2.0 - k * (tx * tx + ux * ux), m01 = -k * (tx * ty + ux * uy), m02 = -k * (tx * tz + ux * uz);
                const double m11 = 2.0 - k * (ty * ty + uy * uy), m12 = -k * (ty * tz + uy * uz), m22 = 2.0 - k * (tz * tz + uz * uz);
                // inverse by cofactors over the determinant (what Eigen's 3 x 3 inverse does; no singularity check there either)
                const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
                const double idet = 1.0 / (m00 * c00 + m01 * c01 + m02 * c02);
                double I[3][3];
                I[0][0] = c00 * idet; I[0][1] = c01 * idet; I[0][2] = c02 * idet;
                I[1][1] = (m00 * m22 - m02 * m02) * idet; I[1][2] = (m01 * m02 - m00 * m12) * idet;
                I[2][2] = (m00 * m11 - m01 * m01) * idet;
                I[1][0] = I[0][1]; I[2][0] = I[0][2]; I[2][1] = I[1][2];
                const double p3[3] = {(double)dpx, (double)dpy, (double)dpz};
                const double dd[3] = {(double)(dpx - dqx), (double)(dpy - dqy), (double)(dpz - dqz)};
                // columns of A: a0 = (0, -pz, py), a1 = (pz, 0, -px), a2 = (-py, px, 0), a3..5 = e0..2;  h_v = M^-1 a_v
                double Acol[6][3] = {{0.0, -p3[2], p3[1]}, {p3[2], 0.0, -p3[0]}, {-p3[1], p3[0], 0.0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
                double H[6][3];
#pragma unroll
                for (int v = 0; v < 6; ++v)
#pragma unroll
                    for (int c = 0; c < 3; ++c) H[v][c] = I[c][0] * Acol[v][0] + I[c][1] * Acol[v][1] + I[c][2] * Acol[v][2];
                int kk = 2;
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int v = u; v < 6; ++v)
                        acc[kk++] += (A)(Acol[u][0] * H[v][0] + Acol[u][1] * H[v][1] + Acol[u][2] * H[v][2]);
#pragma unroll
                for (int u = 0; u < 6; ++u) acc[23 + u] += (A)(H[u][0] * dd[0] + H[u][1] * dd[1] + H[u][2] * dd[2]);
            } else {
                A nx, ny, nz;  // (float32 normals, or the doubles Open3D keeps: F4L_ICP_NORMALS_F64)
                if (a.normals_f64) {
                    const double *nn = reinterpret_cast<const double *>(a.tgt_normals) + 3 * (t0 + bj);
                    nx = (A)nn[0]; ny = (A)nn[1]; nz = (A)nn[2];
                } else {
                    const float *nn = a.tgt_normals + 3 * (t0 + bj);
                    nx = nn[0]; ny = nn[1]; nz = nn[2];
                }
                const A r = (dpx - dqx) * nx + (dpy - dqy) * ny + (dpz - dqz) * nz;
                A J[6];
                J[0] = dpy * nz - dpz * ny; J[1] = dpz * nx - dpx * nz; J[2] = dpx * ny - dpy * nx;
                J[3] = nx; J[4] = ny; J[5] = nz;
                int k = 2;
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int v = u; v < 6; ++v) acc[k++] += J[u] * J[v];  // 21 upper-triangular terms
#pragma unroll
                for (int u = 0; u < 6; ++u) acc[23 + u] += J[u] * r;
            }
        };

        // The two phases, once for each way of summing.  DEFER (point-to-plane with certificates, round 6): the correspondences are
        // only FOUND here -- every source point's is in `prev` afterwards -- and summed in a loop of their own below.  The plane
        // estimator carries 29 double sums per lane (58 registers): live across the search they left the bulk shape with 976 bytes
        // of scratch per lane and ~170 reloads per batch of 64 points (46 ms at C4 against 18.5 for point-to-point); summed where
        // nothing else is live they stay in registers.  The sums meet other lanes in another order: the same trajectories to
        // rounding (tests: the shapes against each other and the oracle).
        // (Only the throughput shapes' translation units take this form: as a generic lambda the sweep schedules 1 % better there under the
        //  default scheduler and 2-4 % worse in the shapes a 1 M-point tile runs in -- C2 0.688 against 0.660 ms -- which keep the plain block.)
        ICP_SWEEP_OPEN
            int n_search = ns;  // source points that need a search in this pass
            if (use_cert) {
                // ---- phase 1: re-measure last pass's correspondence of every source point; certify or queue
                unsigned short *myq = queue + wave * seg;
                int nq = 0;
                for (int base = 0; base < ns; base += NT) {
                    const int i = base + tid;
                    const bool valid = i < ns;
                    const int ii = valid ? i : ns - 1;  // idle lanes recompute the last point (keeps loads in bounds)
                    F x, y, z;
                    if (src_in_lds) { x = sl[3 * ii]; y = sl[3 * ii + 1]; z = sl[3 * ii + 2]; }
                    else {
                        // sources that are not staged in LDS come from global memory one batch ahead: (nfx, nfy, nfz) was
                        // requested while the previous batch (or, for the first batch, the previous pass) was at work
                        const float cx_ = nfx, cy_ = nfy, cz_ = nfz;
                        int in = base + NT + tid;  // this thread's point in the next batch; past the end: the first batch again
                        in = base + NT < ns ? (in < ns ? in : ns - 1) : (tid < ns ? tid : ns - 1);
                        nfx = sg[3 * in]; nfy = sg[3 * in + 1]; nfz = sg[3 * in + 2];
                        x = (F)cx_ - (F)ox; y = (F)cy_ - (F)oy; z = (F)cz_ - (F)oz;
                    }
                    const F px = R0 * x + R1 * y + R2 * z + t0f;
                    const F py = R3 * x + R4 * y + R5 * z + t1f;
                    const F pz = R6 * x + R7 * y + R8 * z + t2f;
                    const int pv = (int)prev[ii];  // 0xffff: never searched, 0xfffe: nothing within the search radius
                    // distance every OTHER target is still known to keep.  The certificate arrays are float32 in both modes:
                    // bounds stored rounded down, positions with an allowance for their rounding.
                    F room = (F)mabs[ii] - dsum;
                    if (per_point) {
                        const F mx = px - (F)ps[3 * ii], my = py - (F)ps[3 * ii + 1], mz = pz - (F)ps[3 * ii + 2];
                        F moved = grid_sqrt<F>(grid_d2(mx, my, mz)) * (F)1.000001;
                        if (sizeof(F) == 8) moved += (F)2e-7 * (fabs(px) + fabs(py) + fabs(pz));
                        room = (F)mabs[ii] - moved;
                    }
                    const GridPt<F> q = tl[pv < 0xfffe ? pv : 0];
                    F qx, qy, qz;
                    grid_rel(g, q, qx, qy, qz);
                    // (the same expression the search evaluates: certified and searched distances are the same bits)
                    const F d = grid_d2(grid_query(px, g.ox) - grid_coord(q.x, px), grid_query(py, g.oy) - grid_coord(q.y, py),
                                        grid_query(pz, g.oz) - grid_coord(q.z, pz));
                    bool cert = pv < 0xfffe ? grid_sqrt<F>(d) * (F)1.000001 < room : (pv == 0xfffe && room > rF * (F)1.000001);
                    cert = cert && valid;
                    const bool hit = cert && pv < 0xfffe && d < r2;
                    const int qid = (int)(q.tag >> 16);
                    if (!DEFER && hit) accumulate(acc, px, py, pz, qx, qy, qz, d, qid, ii);
                    if (a.corr_out && cert) a.corr_out[s0 + i] = hit ? qid : -1;
                    const bool need = valid && !cert;
                    const unsigned long long m = __ballot(need);
                    if (need)  // position among the wave's queued lanes: set bits of m below this lane
                        myq[nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] =
                            (unsigned short)i;
                    nq += __builtin_popcountll(m);
                }
                if (lane == 0) qcnt[wave] = nq;
                __syncthreads();
                n_search = 0;
    #pragma unroll
                for (int w = 0; w < NW; ++w) n_search += qcnt[w];
            }
            PROF_T(pt_p0b);
            PROF_ADD(15, pt_p0b, pt_p0);

            // ---- phase 2: search the queued points (all points without certificates), 64 per wave
            for (int base = wave * 64; base < n_search; base += NT) {  // a wave without queued points skips the batch
                const int k = base + lane;
                const bool valid = k < n_search;
                int i = valid ? k : n_search - 1;
                if (use_cert) {
                    int w = 0, loc = i;
    #pragma unroll
                    for (int u = 0; u < NW - 1; ++u) {
                        const int c = qcnt[u];
                        const bool beyond = (w == u) && loc >= c;
                        loc = beyond ? loc - c : loc;
                        w = beyond ? w + 1 : w;
                    }
                    i = (int)queue[w * seg + loc];
                }
                F x, y, z;
                if (src_in_lds) { x = sl[3 * i]; y = sl[3 * i + 1]; z = sl[3 * i + 2]; }
                else { x = (F)sg[3 * i] - (F)ox; y = (F)sg[3 * i + 1] - (F)oy; z = (F)sg[3 * i + 2] - (F)oz; }
                const F px = R0 * x + R1 * y + R2 * z + t0f;
                const F py = R3 * x + R4 * y + R5 * z + t1f;
                const F pz = R6 * x + R7 * y + R8 * z + t2f;
                Best<F> best;
                F b0 = rs2;
                if (tgt_in_lds) {
                    if (use_cert) {
                        // last pass's correspondence, re-measured, bounds the search from the start
                        const int pv = (int)prev[i];
                        if (pv < 0xfffe && !(a.debug & 8)) {
                            const GridPt<F> q = tl[pv];
                            const F bb = grid_sqrt<F>(grid_d2(grid_query(px, g.ox) - grid_coord(q.x, px), grid_query(py, g.oy) - grid_coord(q.y, py),
                                                              grid_query(pz, g.oz) - grid_coord(q.z, pz))) * (F)1.000001 + mu;
                            b0 = bb * bb < rs2 ? bb * bb : rs2;
                        }
                    }
                    best.init(b0);
    #ifdef F4L_ICP_PROF
                    grid_nn<F, NT>(g, tl, nt, E, rl, valid, px, py, pz, best,
                                   ((a.debug & 64) && ns <= a.prof_max_n && nt <= a.prof_max_n) ? a.prof : nullptr);
    #else
                    grid_nn<F, NT>(g, tl, nt, E, rl, valid, px, py, pz, best);
    #endif
                    if (use_cert && valid) {
                        const F m2 = best.second < b0 ? best.second : b0;
                        prev[i] = (unsigned short)(best.found() ? best.slot() : 0xfffe);
                        if (per_point) {
                            mabs[i] = (float)(grid_sqrt<F>(m2) * (F)0.999999);
                            ps[3 * i] = (float)px; ps[3 * i + 1] = (float)py; ps[3 * i + 2] = (float)pz;
                        } else mabs[i] = (float)((grid_sqrt<F>(m2) * (F)0.999999 + dsum) * (F)0.9999998);
                    }
                } else {
                    best.init(r2);
                    nn_global<F>(tg, nt, ox, oy, oz, px, py, pz, best);
                }
                const bool hit = valid && best.found() && best.d2() < r2;  // SearchHybrid: d2 < r^2
                const int bj = tgt_in_lds ? best.id() : (int)best.tag();  // index inside the target patch
                if (a.corr_out && valid) a.corr_out[s0 + i] = hit ? bj : -1;
                if (!DEFER && hit) {
                    F qx, qy, qz;
                    if (tgt_in_lds) { const GridPt<F> q = tl[best.slot()]; grid_rel(g, q, qx, qy, qz); }
                    else { qx = (F)tg[3 * bj] - (F)ox; qy = (F)tg[3 * bj + 1] - (F)oy; qz = (F)tg[3 * bj + 2] - (F)oz; }
                    accumulate(acc, px, py, pz, qx, qy, qz, best.d2(), bj, i);
                }
            }

            if constexpr (DEFER) {
                __syncthreads();  // (a searched point's correspondence was written by the lane that searched it)
                A sums[NV];        // (the pass's sums start to exist here: nothing of them is live across the search)
#pragma unroll
                for (int v = 0; v < NV; ++v) sums[v] = (A)0;
                for (int base = 0; base < ns; base += NT) {
                    const int i = base + tid;
                    const bool valid = i < ns;
                    const int ii = valid ? i : ns - 1;
                    F x, y, z;
                    if (src_in_lds) { x = sl[3 * ii]; y = sl[3 * ii + 1]; z = sl[3 * ii + 2]; }
                    else { x = (F)sg[3 * ii] - (F)ox; y = (F)sg[3 * ii + 1] - (F)oy; z = (F)sg[3 * ii + 2] - (F)oz; }
                    const F px = R0 * x + R1 * y + R2 * z + t0f;
                    const F py = R3 * x + R4 * y + R5 * z + t1f;
                    const F pz = R6 * x + R7 * y + R8 * z + t2f;
                    const int pv = (int)prev[ii];
                    const GridPt<F> q = tl[pv < 0xfffe ? pv : 0];
                    F qx, qy, qz;
                    grid_rel(g, q, qx, qy, qz);
                    // (the expression of phase 1 and of the search: the same bits, so the same `d < r2`)
                    const F d = grid_d2(grid_query(px, g.ox) - grid_coord(q.x, px), grid_query(py, g.oy) - grid_coord(q.y, py),
                                        grid_query(pz, g.oz) - grid_coord(q.z, pz));
                    if (valid && pv < 0xfffe && d < r2) accumulate(sums, px, py, pz, qx, qy, qz, d, (int)(q.tag >> 16), ii);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = sums[v];
            }
#ifdef F4L_ICP_PROF
            PROF_T(pt_p1_);
            PROF_ADD(2, pt_p1_, pt_p0b);
#endif
        ICP_SWEEP_CLOSE

        PROF_T(pt_p1);
        // DPP reduction inside the wave, then the NW partials (as double) through LDS; one wave solves
        {
            A xs[NV / 4], ys[NV % 4 > 0 ? NV % 4 : 1];
            row_sums_transposed<NV, A>(acc, xs, ys);
            if ((lane & 12) == 0) {  // the first quad of every row writes the row's sums (as double)
                double *row = scratch + (wave * 4 + (lane >> 4)) * NV;
#pragma unroll
                for (int m = 0; m < NV / 4; ++m) row[4 * m + (lane & 3)] = (double)xs[m];
                if ((lane & 3) == 0) {
#pragma unroll
                    for (int j = 0; j < NV % 4; ++j) row[4 * (NV / 4) + j] = (double)ys[j];
                }
            }
        }
        PROF_T(pt_p1b);
        PROF_ADD(12, pt_p1b, pt_p1);
        __syncthreads();
        PROF_T(pt_p2);
        PROF_ADD(3, pt_p2, pt_p1);
        if (wave == solver) {
            // The solve is one long dependent chain executed by a single wave while the rest of the workgroup
            // waits: let it win the issue arbitration against the other workgroups' waves on this SIMD.
            __builtin_amdgcn_s_setprio(3);
            double tot[NV];
            {
                // lane i sums the partial rows of value i (one LDS read per row), then the totals go to scalar
                // registers: keeps SUM_ROWS * NV partial sums from being live at once
                const int vi = lane < NV ? lane : NV - 1;
                double t = scratch[vi];
#pragma unroll
                for (int w = 1; w < SUM_ROWS; ++w) t += scratch[w * NV + vi];
                const long long tb = __double_as_longlong(t);
                const int tlo = (int)(tb & 0xffffffffLL), thi = (int)(tb >> 32);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int lo = __builtin_amdgcn_readlane(tlo, i), hi = __builtin_amdgcn_readlane(thi, i);
                    tot[i] = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                }
            }
            const double fitness = state[13], rmse = state[14];
            int iters = (int)state[15];
            const double m = tot[0];
            const double fit_new = m > 0.0 ? m / (double)ns : 0.0;
            const double rmse_new = m > 0.0 ? sqrt(tot[1] / m) : 0.0;
            bool done = false;
            if (pass > 0) {
                iters = pass;
                if (!a.fixed_iters && fabs(fitness - fit_new) < a.rel_fitness && fabs(rmse - rmse_new) < a.rel_rmse)
                    done = true;
            }
            if (pass == a.max_iter) done = true;
            // update [Ru | tu] of this iteration (origin-relative): p_new = Ru p_old + tu
            double Ru[9], tu[3];
            bool have = false, bad_step = false;
            if (!done && m > 0.0) {
                have = true;
                if (MODE == F4L_ICP_POINT2POINT) {
                    // Eigen::umeyama without scaling.  Means about the centring point (zero shift when the sums are
                    // uncentred); the covariance is shift invariant.
                    const double im = fast_rcp(m);
                    double sg9[9];
                    {
                        const double cm0 = tot[2] * im, cm1 = tot[3] * im, cm2 = tot[4] * im;
                        const double cq0 = tot[5] * im, cq1 = tot[6] * im, cq2 = tot[7] * im;
                        sg9[0] = tot[8] * im - cq0 * cm0; sg9[1] = tot[9] * im - cq0 * cm1; sg9[2] = tot[10] * im - cq0 * cm2;
                        sg9[3] = tot[11] * im - cq1 * cm0; sg9[4] = tot[12] * im - cq1 * cm1; sg9[5] = tot[13] * im - cq1 * cm2;
                        sg9[6] = tot[14] * im - cq2 * cm0; sg9[7] = tot[15] * im - cq2 * cm1; sg9[8] = tot[16] * im - cq2 * cm2;
                    }
                    // nearly aligned clouds (every iteration but possibly the first): Newton on SO(3); otherwise,
                    // and for rank-deficient sums, the warm-started Jacobi SVD (same optimum, U diag(1,1,det) V^T)
                    if ((a.debug & 128) || !rot_newton(sg9, Ru)) {
                        double U[9], V[9], V0[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) V0[i] = state[16 + i];
                        svd3_warm(sg9, V0, U, V);
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < 9; ++i) state[16 + i] = V[i];
                        }
                        const double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
                        mul_diag_bt(U, sgn, V, Ru);
                    }
                    // (the means are recomputed rather than kept alive across the rotation solve)
                    // the shift the lanes applied: the float32 value of the centroid image
                    const double sh0 = CENTRED ? (double)(F)state[30] : 0.0, sh1 = CENTRED ? (double)(F)state[31] : 0.0,
                                 sh2 = CENTRED ? (double)(F)state[32] : 0.0;
                    const double mp0 = tot[2] * im + sh0, mp1 = tot[3] * im + sh1, mp2 = tot[4] * im + sh2;
                    const double mq0 = tot[5] * im + sh0, mq1 = tot[6] * im + sh1, mq2 = tot[7] * im + sh2;
                    tu[0] = mq0 - (Ru[0] * mp0 + Ru[1] * mp1 + Ru[2] * mp2);
                    tu[1] = mq1 - (Ru[3] * mp0 + Ru[4] * mp1 + Ru[5] * mp2);
                    tu[2] = mq2 - (Ru[6] * mp0 + Ru[7] * mp1 + Ru[8] * mp2);
                } else {
                    double x[6];
                    bool caller_frame = false;  // x is the solution about the caller's origin (else about the patch's)
                    if (MODE == F4L_ICP_GENERALIZED || a.p2pl_open3d) {
                        // Open3D's own step (SolveJacobianSystemAndObtainExtrinsicMatrix; the only one generalized ICP has): solved whatever the rank of the system, applied whenever there is a correspondence
                        p2plane_step_open3d((int)((const unsigned char *)scratch - smem_raw), SUM_ROWS,
                                            (int)((const unsigned char *)state - smem_raw), ox, oy, oz);
                        have = true;
                        caller_frame = true;
#pragma unroll
                        for (int u = 0; u < 6; ++u) { x[u] = state[40 + u]; have = have && isfinite(x[u]); }
                        if constexpr (MODE == F4L_ICP_GENERALIZED) {  // (a NaN diagonal reads as "all zero" to the ldlt restatement: x = 0, finite)
                            double chk = 0.0;
#pragma unroll
                            for (int u = 2; u < NV; ++u) chk += tot[u];
                            have = have && isfinite(chk);
                        }
                        // A step that is not finite (generalized ICP with epsilon = 0 -- the reference's call -- on a pair of exactly
                        // parallel normals: M is singular, 1 / det = inf, and the NaN is in every sum of the pass) would be Open3D's
                        // NaN transform, a visible failure.  Here the patch keeps its last finite transform, stops, and says so:
                        // iters = -2 (include/f4l.h).
                        bad_step = !have;
                    } else {
                        double M[6][7];
                        int k = 2;
#pragma unroll
                        for (int u = 0; u < 6; ++u)
#pragma unroll
                            for (int v = u; v < 6; ++v) { M[u][v] = tot[k]; M[v][u] = tot[k]; ++k; }
#pragma unroll
                        for (int u = 0; u < 6; ++u) M[u][6] = -tot[23 + u];
                        // (fewer than six correspondences cannot pin six unknowns: no step, the transform stays and the loop
                        //  ends on its criteria -- the ROBUST rule of include/f4l.h; Open3D's own is the branch above)
                        have = m >= 6.0 && ((a.debug & 256) ? solve6(M, x) : solve6_spd(M, x));
                    }
                    if (have) {
                        // o3d TransformVector6dToMatrix4d: Rz(x2) Ry(x1) Rx(x0), translation x[3:6]
                        double ca, sa, cb, sb, cg, sgm;
                        sincos_step(x[0], sa, ca);
                        sincos_step(x[1], sb, cb);
                        sincos_step(x[2], sgm, cg);
                        Ru[0] = cg * cb; Ru[1] = cg * sb * sa - sgm * ca; Ru[2] = cg * sb * ca + sgm * sa;
                        Ru[3] = sgm * cb; Ru[4] = sgm * sb * sa + cg * ca; Ru[5] = sgm * sb * ca - cg * sa;
                        Ru[6] = -sb; Ru[7] = cb * sa; Ru[8] = cb * ca;
                        // the system was built about the patch origin o: t' = t + x cross o.  Undo that, then express
                        // the reference's update [Rot(x) | t] (a rotation about the GLOBAL origin) in origin-relative
                        // coordinates: tu = Rot(x) o + t - o.
                        const double o0 = ox, o1 = oy, o2 = oz;
                        const double tg0 = caller_frame ? x[3] : x[3] - (x[1] * o2 - x[2] * o1);
                        const double tg1 = caller_frame ? x[4] : x[4] - (x[2] * o0 - x[0] * o2);
                        const double tg2 = caller_frame ? x[5] : x[5] - (x[0] * o1 - x[1] * o0);
                        tu[0] = Ru[0] * o0 + Ru[1] * o1 + Ru[2] * o2 + tg0 - o0;
                        tu[1] = Ru[3] * o0 + Ru[4] * o1 + Ru[5] * o2 + tg1 - o1;
                        tu[2] = Ru[6] * o0 + Ru[7] * o1 + Ru[8] * o2 + tg2 - o2;
                    }
                }
            }
            if (lane == 0) {
                state[12] = (done || bad_step) ? 1.0 : 0.0;
                state[13] = fit_new; state[14] = rmse_new; state[15] = bad_step ? -2.0 : (double)iters;
            }
            if (have) {  // T <- update * T; the running transform is only now fetched from LDS
                const double cp0 = state[30], cp1 = state[31], cp2 = state[32];  // centroid image under the old T
                // bound on how far any source point moves with this update (see the kernel's header comment):
                // rotation about the patch centroid's image times the patch radius, plus the centroid's own step
                double fro = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { const double e = Ru[i] - ((i % 4 == 0) ? 1.0 : 0.0); fro += e * e; }
                const double n0 = Ru[0] * cp0 + Ru[1] * cp1 + Ru[2] * cp2 + tu[0];  // centroid image under the new T
                const double n1 = Ru[3] * cp0 + Ru[4] * cp1 + Ru[5] * cp2 + tu[1];
                const double n2 = Ru[6] * cp0 + Ru[7] * cp1 + Ru[8] * cp2 + tu[2];
                const double m0 = n0 - cp0, m1 = n1 - cp1, m2 = n2 - cp2;
                const double cpn = fast_sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
                // positions are evaluated in F from the rounded transform: a few ulps of their magnitude
                const double eps_pos = sizeof(F) == 4 ? 4e-6 : 1e-14;
                double motion = fast_sqrt(fro) * state[28] + fast_sqrt(m0 * m0 + m1 * m1 + m2 * m2) + eps_pos * (state[28] + cpn);
                motion *= 1.0 + 1e-9;
                double Rc[9], tc[3], Rn[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rc[i] = state[i];
                tc[0] = state[9]; tc[1] = state[10]; tc[2] = state[11];
                mul3(Ru, Rc, Rn);
                const double tn0 = Ru[0] * tc[0] + Ru[1] * tc[1] + Ru[2] * tc[2] + tu[0];
                const double tn1 = Ru[3] * tc[0] + Ru[4] * tc[1] + Ru[5] * tc[2] + tu[1];
                const double tn2 = Ru[6] * tc[0] + Ru[7] * tc[1] + Ru[8] * tc[2] + tu[2];
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) state[i] = Rn[i];
                    state[9] = tn0; state[10] = tn1; state[11] = tn2;
                    state[29] += motion;
                    state[30] = n0; state[31] = n1; state[32] = n2;
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }

        PROF_T(pt_p3);
        PROF_ADD(4, pt_p3, pt_p2);
        __syncthreads();
        PROF_T(pt_p4);
        PROF_ADD(5, pt_p4, pt_p3);
        const bool finished = state[12] != 0.0;
        if (finished) break;  // uniform across the workgroup
    }

    PROF_T(pt_end);
    PROF_ADD(0, pt_end, pt_start);
#ifdef F4L_ICP_PROF
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    prof_acc[14] += rt_end - rt_start;
    if (a.prof && tid == 0) { a.prof[32 + 2 * p] = rt_start; a.prof[33 + 2 * p] = rt_end; }
    if (a.prof && tid == 0 && ns <= a.prof_max_n && nt <= a.prof_max_n) {  // (F4L_ICP_PROF_MAXN: small patches only)
        const int slots[9] = {0, 1, 2, 3, 4, 5, 15, 12, 14};
        for (int i = 0; i < 9; ++i) atomicAdd(&a.prof[slots[i]], prof_acc[slots[i]]);
        atomicAdd(&a.prof[26], 1ULL);
    }
#endif
    if (tid == 0) {
        // back to global coordinates: t = tc - Rc o + o
        const double o0 = ox, o1 = oy, o2 = oz;
        double Rc[9], tc[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rc[i] = state[i];
        tc[0] = state[9]; tc[1] = state[10]; tc[2] = state[11];
        const double fitness = state[13], rmse = state[14];
        const int iters = skipped ? -1 : (int)state[15];
        double *T = a.T_out + 16 * p;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T[4 * i] = Rc[3 * i]; T[4 * i + 1] = Rc[3 * i + 1]; T[4 * i + 2] = Rc[3 * i + 2];
        }
        T[3] = tc[0] - (Rc[0] * o0 + Rc[1] * o1 + Rc[2] * o2) + o0;
        T[7] = tc[1] - (Rc[3] * o0 + Rc[4] * o1 + Rc[5] * o2) + o1;
        T[11] = tc[2] - (Rc[6] * o0 + Rc[7] * o1 + Rc[8] * o2) + o2;
        T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
        if (a.fitness_out) a.fitness_out[p] = fitness;
        if (a.rmse_out) a.rmse_out[p] = rmse;
        if (a.iters_out) a.iters_out[p] = iters;
    }
    if (!active && a.corr_out)
        for (int i = tid; i < ns; i += NT) a.corr_out[s0 + i] = -1;
    if (a.rows_out && !skipped) {
        // fused displacement rows [s, T s] (src/coarse_to_fine_matching_base.py:3371-3374,3408): the arithmetic of
        // apply_transform_kernel on the global 4x4 this thread rebuilds from the final LDS state
        const double o0 = ox, o1 = oy, o2 = oz;
        double r[9], tr[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = state[i];
        tr[0] = state[9] - (r[0] * o0 + r[1] * o1 + r[2] * o2) + o0;
        tr[1] = state[10] - (r[3] * o0 + r[4] * o1 + r[5] * o2) + o1;
        tr[2] = state[11] - (r[6] * o0 + r[7] * o1 + r[8] * o2) + o2;
        const int64_t w0 = a.rows_off ? a.rows_off[p] : s0;
        const int nrow = a.rows_off ? (int)(a.rows_off[p + 1] - w0) : ns;
        const float *__restrict__ wg = a.rows_src ? a.rows_src + 3 * w0 : sg;
        float *__restrict__ out6 = a.rows_out + 6 * w0;
        for (int i = tid; i < nrow; i += NT) {
            const float xf = wg[3 * i], yf = wg[3 * i + 1], zf = wg[3 * i + 2];
            const double x = xf, y = yf, z = zf;
            float *o6 = out6 + 6 * i;
            o6[0] = xf; o6[1] = yf; o6[2] = zf;
            o6[3] = (float)(r[0] * x + r[1] * y + r[2] * z + tr[0]);
            o6[4] = (float)(r[3] * x + r[4] * y + r[5] * z + tr[1]);
            o6[5] = (float)(r[6] * x + r[7] * y + r[8] * z + tr[2]);
        }
    }
}

}  // namespace f4l

// The throughput shapes (WIDE: the bulk class of a large batch, bench.py's headline launch) live in translation units of their own,
// this file compiled again with -DF4L_ICP_BULK_TU=1 (point-to-point) and -DF4L_ICP_BULK_TU=2 (point-to-plane): the instruction
// scheduler is a per-file compiler option and the shapes want different ones (csrc/Makefile: icp_bulk.o, icp_bulk_plane.o;
// profiles/r6_aa_icp_bulk_schedulers.log).
namespace f4l {
#ifdef F4L_ICP_BULK_TU
#if F4L_ICP_BULK_TU == 2  // (icp_bulk_plane.o: the plane estimator's shape, under the scheduler that suits it -- csrc/Makefile)
template __global__ void icp_kernel<1, 2, double, true>(IcpArgs);
#else
template __global__ void icp_kernel<0, 2, double, true>(IcpArgs);
#endif
#else
extern template __global__ void icp_kernel<0, 2, double, true>(IcpArgs);
extern template __global__ void icp_kernel<1, 2, double, true>(IcpArgs);
#endif
}  // namespace f4l

#ifndef F4L_ICP_BULK_TU
#include "icp_rows.h"

namespace f4l {

// Size classes: patch p goes to the first class whose bound holds max(sources, targets) of the patch.  The order inside
// a class list depends on scheduling; nothing downstream does (every patch is solved on its own).
constexpr int ICP_MAX_CLASSES = 12;
struct ClassBounds { int n; int bound[ICP_MAX_CLASSES]; };
__global__ void icp_bin_patches(const int64_t *__restrict__ src_off, const int64_t *__restrict__ tgt_off, int P,
                                ClassBounds cb, int *__restrict__ cnt, int *__restrict__ list) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = (int)(threadIdx.x & 63);
    int k = -1;
    if (p < P) {
        const int64_t ns = src_off[p + 1] - src_off[p], nt = tgt_off[p + 1] - tgt_off[p];
        const int64_t m = ns > nt ? ns : nt;
        k = 0;
        while (k < cb.n - 1 && m > cb.bound[k]) ++k;
    }
    for (int c = 0; c < cb.n; ++c) {  // one atomic per wave and class
        const unsigned long long m = __ballot(k == c);
        if (m == 0ULL) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&cnt[c], __builtin_popcountll(m));
        base = __shfl(base, leader, 64);
        if (k == c) list[(size_t)c * P + base + __builtin_popcountll(m & ((1ULL << lane) - 1ULL))] = p;
    }
}

template <int MODE, int NW, typename F, bool WIDE = false>
static int launch_icp_one(const IcpArgs &a, size_t lds, hipStream_t st) {
    // (measurement only: F4L_ICP_LDS_PAD = bytes of LDS asked for and never used -- fewer workgroups per CU, nothing else changes:
    //  what ONE workgroup per CU is worth, profiles/r5_*_occupancy_slope*.log)
    if (const char *e = getenv("F4L_ICP_LDS_PAD")) lds += (size_t)atoi(e);
    if (lds > 64 * 1024)  // opt in to > 64 KiB of dynamic LDS
        F4L_HIP_CHECK(hipFuncSetAttribute((const void *)icp_kernel<MODE, NW, F, WIDE>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((icp_kernel<MODE, NW, F, WIDE>), dim3((unsigned)a.P), dim3(NW * 64), lds, st, a);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

template <typename F>
static int launch_icp(const IcpArgs &a, int mode, int nw, size_t lds, hipStream_t st, bool wide = false) {
    // (measured and not kept, round 6: a second kernel for point-to-point / float64 patches that keeps the correspondence sums LAZILY --
    //  moments of the counted pairs in original coordinates, per-point deadlines in units of the patch-wide motion bound, only status
    //  changes touch the sums: tools/experiments/icp_lazy.h.  C4 18.25 against 18.25 ms, C2 0.80 against 0.69, C3 17.7 against 15.0:
    //  the benchmark's patches never go quiet -- 8 % of a patch's points are still searched in pass 20, profiles/r4_f_searches_per_pass.log
    //  -- so the sweep it saves is paid back by the bookkeeping of the points that do change.)
    if (mode == F4L_ICP_POINT2POINT) {
        if (nw == 1) return launch_icp_one<0, 1, F>(a, lds, st);
        if (nw == 2) return wide ? launch_icp_one<0, 2, F, true>(a, lds, st) : launch_icp_one<0, 2, F>(a, lds, st);
        return launch_icp_one<0, 4, F>(a, lds, st);
    }
    if (mode == F4L_ICP_GENERALIZED) {  // double only, no wide shape (icp_plan)
        if constexpr (sizeof(F) == 8) {
            if (nw == 1) return launch_icp_one<2, 1, F>(a, lds, st);
            if (nw == 2) return launch_icp_one<2, 2, F>(a, lds, st);
            return launch_icp_one<2, 4, F>(a, lds, st);
        } else return F4L_EUNSUPPORTED;
    }
    if (nw == 1) return launch_icp_one<1, 1, F>(a, lds, st);
    if (nw == 2) return wide ? launch_icp_one<1, 2, F, true>(a, lds, st) : launch_icp_one<1, 2, F>(a, lds, st);
    return launch_icp_one<1, 4, F>(a, lds, st);
}

template <typename F>
static int launch_rows(const IcpArgs &a, int lp, hipStream_t st) {
    const int per_block = ROWS_WAVES * (64 / lp);
    const unsigned blocks = (unsigned)((a.P + per_block - 1) / per_block);
    if (lp == 16) hipLaunchKernelGGL((icp_rows_kernel<F, 16>), dim3(blocks), dim3(ROWS_WAVES * 64), 0, st, a);
    else hipLaunchKernelGGL((icp_rows_kernel<F, 32>), dim3(blocks), dim3(ROWS_WAVES * 64), 0, st, a);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

static inline int pow2_ceil(int64_t v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}
}  // namespace f4l

namespace f4l {
// Streams of the library's own for size classes that run beside the caller's stream; created once per device, never
// destroyed (a process keeps its device for life).  Non-blocking: they synchronise with the caller's stream through
// events only.
static hipStream_t class_stream(int which) {
    constexpr int MAX_DEV = 16;
    static std::mutex mu;
    static hipStream_t pool[MAX_DEV][ICP_MAX_CLASSES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV || which < 0 || which >= ICP_MAX_CLASSES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!pool[dev][which] && hipStreamCreateWithFlags(&pool[dev][which], hipStreamNonBlocking) != hipSuccess) return nullptr;
    return pool[dev][which];
}

// Workgroup shape and LDS layout for patches of at most max_src sources and max_tgt targets.
struct IcpPlan { int nw, tgt_cap, cell_cap, cert_cap, src_cap, pp_cap; size_t lds; bool wide; int rows_lp; };
// `throughput`: the launch holds many rounds of workgroups (see icp_launch_host): the shape that moves the most patches per
// second wins, not the one that finishes a single patch soonest.
static IcpPlan icp_plan(int64_t max_src_patch_host, int64_t max_tgt_patch_host, bool f64, int mode, bool throughput = false,
                        bool rows_ok = false) {
    const size_t pt = sizeof(GridPt<float>);  // 16 B in both modes
    // Waves per patch.  A patch's pass is a chain (certify, search, reduce, solve, three barriers): four waves finish it
    // soonest, which is what counts while a launch is only a few rounds of workgroups (C2: 0.72 ms against 0.80 ms with two
    // waves).  In the throughput regime two waves with the register budget of three waves per SIMD (no scratch) and five to
    // six workgroups per CU move more patches: C4 22.9 instead of 26.1 ms (measured grid: DESIGN.md section 5).  Patches
    // that fit one or two wavefronts get just those.
    int nw = 4;
    bool wide = false;
    if (max_src_patch_host <= 64) nw = 1;
    else if (max_src_patch_host <= 128) nw = 2;
    else if (throughput && mode != F4L_ICP_GENERALIZED && !getenv("F4L_ICP_NOWIDE")) { nw = 2; wide = f64; }  // (the float32 build fits 128 VGPRs without scratch)
    const bool tiers = throughput && nw == 2;
    { const char *e = getenv("F4L_ICP_WAVES"); if (e) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) { nw = v; wide = wide && v == 2; } } }
    if (getenv("F4L_ICP_WIDE")) wide = nw == 2;
    if (mode == F4L_ICP_GENERALIZED) wide = false;

    // LDS plan: targets first (they make the grid possible), then the prefix table, then the certificate arrays
    // (partial-sum region: as SCRATCH in icp_kernel)
    const int nv = mode == F4L_ICP_POINT2POINT ? 17 : 29;
    const int sum_doubles = (4 * nw * nv + 1) & ~1;
    const size_t fixed = (size_t)(sum_doubles + 48) * sizeof(double) + 16 + (size_t)(GRID_ROWS + 1) * nw * 64 * sizeof(unsigned int);
    int tgt_cap = (int)(max_tgt_patch_host < ICP_TGT_MAX ? max_tgt_patch_host : ICP_TGT_MAX);
    if (tgt_cap < 1) tgt_cap = 1;
    int cell_cap = (int)((2 * (int64_t)tgt_cap + 255) & ~(int64_t)255);  // ~2 cells per target point
    cell_cap = cell_cap < 512 ? 512 : (cell_cap > ICP_CELL_MAX ? ICP_CELL_MAX : cell_cap);
    auto table_bytes = [](int cells) { return ((size_t)cells + 8) * 2; };
    while (fixed + (size_t)(tgt_cap + 1) * pt + table_bytes(cell_cap) > (size_t)ICP_LDS_BUDGET && cell_cap > 512) cell_cap >>= 1;
    while (fixed + (size_t)(tgt_cap + 1) * pt + table_bytes(cell_cap) > (size_t)ICP_LDS_BUDGET) tgt_cap -= 256;
    size_t lds = fixed + (size_t)(tgt_cap + 1) * pt + table_bytes(cell_cap);
    const int nt_threads = nw * 64;
    auto cert_bytes = [&](int64_t cap) {
        const size_t seg = (size_t)((cap + nt_threads - 1) / nt_threads) * 64;
        return (size_t)((cap + 3) & ~(int64_t)3) * 4 + (size_t)((cap + 7) & ~(int64_t)7) * 2 + (size_t)nw * seg * 2 + 16;
    };
    int cert_cap = 0;
    if (max_src_patch_host < 0xfff0 && lds + cert_bytes(max_src_patch_host) <= (size_t)ICP_LDS_BUDGET) cert_cap = (int)max_src_patch_host;
    lds += cert_bytes(cert_cap);
    // then, while at least four workgroups still fit a CU (or nothing more than one fits anyway): the per-point
    // certificate positions (worth ~10 % fewer searches than the patch-wide motion bound), then the staged sources
    int src_cap = 0, pp_cap = 0;
    {
        const size_t pb = (size_t)((max_src_patch_host + 3) & ~(int64_t)3) * 3 * 4;
        const size_t sb = (size_t)((max_src_patch_host + 3) & ~(int64_t)3) * 3 * (f64 ? 8 : 4);
        // LDS per workgroup up to which the optional arrays are added: four workgroups per CU (of four waves), or five of two
        // waves at three waves per SIMD
        size_t keep = 40 * 1024;
        if (tiers) {  // six (eight at 128 VGPRs) workgroups of two waves fill the SIMDs: stay in the tier the mandatory arrays reach
            int tier = wide ? 6 : 8;
            while (tier > 1 && lds > (size_t)(160 * 1024) / tier) --tier;
            keep = (size_t)(160 * 1024) / tier;
        }
        if (const char *e = getenv("F4L_ICP_LDS_KEEP")) keep = (size_t)atoi(e) * 1024;
        auto fits = [&](size_t extra) { return lds + extra <= (size_t)ICP_LDS_BUDGET && (lds + extra <= keep || lds > keep); };
        if (cert_cap && !getenv("F4L_ICP_NOPP") && fits(pb)) { pp_cap = cert_cap; lds += pb; }
        if (fits(sb)) { src_cap = (int)max_src_patch_host; lds += sb; }
    }
    lds = (lds + 15) & ~(size_t)15;
    if (getenv("F4L_ICP_PLAN_DEBUG"))
        fprintf(stderr, "[icp plan] max_src %lld max_tgt %lld nw %d: lds %zu B (tgt_cap %d cell_cap %d cert_cap %d pp_cap %d src_cap %d) -> %d workgroups per CU\n",
                (long long)max_src_patch_host, (long long)max_tgt_patch_host, nw, lds, tgt_cap, cell_cap, cert_cap, pp_cap, src_cap, (int)(160 * 1024 / lds));
    IcpPlan pl;
    pl.nw = nw; pl.tgt_cap = tgt_cap; pl.cell_cap = cell_cap; pl.cert_cap = cert_cap; pl.src_cap = src_cap; pl.pp_cap = pp_cap;
    pl.lds = lds;
    pl.wide = wide;
    // small patches, point-to-point: several patches per wave (icp_rows.h) -- 16 lanes per patch up to 64 points, 32 up to 128;
    // where the caller allows it (icp_launch_host: `rows_ok`)
    pl.rows_lp = 0;
#ifndef F4L_ICP_PROF
    if (rows_ok && mode == F4L_ICP_POINT2POINT && !getenv("F4L_ICP_WAVES")) {
        const int64_t m = max_src_patch_host > max_tgt_patch_host ? max_src_patch_host : max_tgt_patch_host;
        pl.rows_lp = m <= 64 ? 16 : (m <= 128 ? 32 : 0);
    }
#endif
    return pl;
}
}  // namespace f4l

namespace f4l {
struct IcpFusedExtra {
    const float *corr_src = nullptr, *corr_ref = nullptr, *corr_w = nullptr;
    const int64_t *corr_off = nullptr;
    double w_thresh = 0.0, eps = 1e-7;
    float *rows_out = nullptr;
    const float *rows_src = nullptr;
    const int64_t *rows_off = nullptr;
    int64_t min_corr = 0;
    int init_round_f32 = 0;
    int normals_f64 = 0;
    int p2pl_open3d = 0;
    const double *src_normals = nullptr;
    double gicp_eps = 0.0;
};
static int icp_launch_host(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                           int64_t P, const double *init_T, const float *tgt_normals, double max_corr_dist,
                           int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                           int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host,
                           int64_t n_src_host, double *T_out, double *fitness_out, double *rmse_out, int32_t *iters_out,
                           int32_t *corr_out, const IcpFusedExtra &fx, void *stream);
}  // namespace f4l

extern "C" int f4l_piecewise_icp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                                 int64_t P, const double *init_T, const float *tgt_normals, double max_corr_dist,
                                 int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                                 int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host,
                                 int64_t n_src_host, double *T_out, double *fitness_out, double *rmse_out,
                                 int32_t *iters_out, int32_t *corr_out, void *stream) {
    f4l::IcpFusedExtra fx;
    fx.normals_f64 = (mode & F4L_ICP_NORMALS_F64) ? 1 : 0;
    fx.p2pl_open3d = (mode & F4L_ICP_P2PL_OPEN3D) ? 1 : 0;
    mode &= ~(F4L_ICP_NORMALS_F64 | F4L_ICP_P2PL_OPEN3D);
    return f4l::icp_launch_host(src, src_off, tgt, tgt_off, P, init_T, tgt_normals, max_corr_dist, max_iter, rel_fitness,
                                rel_rmse, mode, fixed_iters, search_precision, max_src_patch_host, max_tgt_patch_host,
                                n_src_host, T_out, fitness_out, rmse_out, iters_out, corr_out, fx, stream);
}

extern "C" int f4l_piecewise_gicp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                                  const double *init_T, const double *src_normals, const double *tgt_normals, double epsilon,
                                  double max_corr_dist, int max_iter, double rel_fitness, double rel_rmse, int fixed_iters,
                                  int64_t max_src_patch_host, int64_t max_tgt_patch_host, int64_t n_src_host, double *T_out,
                                  double *fitness_out, double *rmse_out, int32_t *iters_out, int32_t *corr_out, void *stream) {
    f4l::IcpFusedExtra fx;
    fx.normals_f64 = 1;
    fx.p2pl_open3d = 1;
    fx.src_normals = src_normals;
    fx.gicp_eps = epsilon;
    return f4l::icp_launch_host(src, src_off, tgt, tgt_off, P, init_T, reinterpret_cast<const float *>(tgt_normals), max_corr_dist,
                                max_iter, rel_fitness, rel_rmse, F4L_ICP_GENERALIZED, fixed_iters, F4L_SEARCH_F64,
                                max_src_patch_host, max_tgt_patch_host, n_src_host, T_out, fitness_out, rmse_out, iters_out,
                                corr_out, fx, stream);
}

extern "C" int f4l_patch_loop(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                              int64_t P, const float *corr_src, const float *corr_ref, const float *corr_w,
                              const int64_t *corr_off, int64_t min_corr, double kabsch_w_thresh, double kabsch_eps,
                              const float *tgt_normals, double max_corr_dist, int max_iter, double rel_fitness,
                              double rel_rmse, int mode, int fixed_iters, int search_precision,
                              int64_t max_src_patch_host, int64_t max_tgt_patch_host, int64_t n_src_host,
                              double *T_out, double *fitness_out, double *rmse_out, int32_t *iters_out,
                              int32_t *corr_out, const float *rows_src, const int64_t *rows_off, float *rows_out,
                              void *stream) {
    if (!corr_off || ((!corr_src || !corr_ref) && P > 0) || min_corr < 0) return F4L_EINVAL;
    if ((rows_src == nullptr) != (rows_off == nullptr)) return F4L_EINVAL;
    f4l::IcpFusedExtra fx;
    fx.corr_src = corr_src; fx.corr_ref = corr_ref; fx.corr_w = corr_w; fx.corr_off = corr_off;
    fx.w_thresh = kabsch_w_thresh; fx.eps = kabsch_eps; fx.rows_out = rows_out;
    fx.rows_src = rows_src; fx.rows_off = rows_off; fx.min_corr = min_corr;
    fx.init_round_f32 = (mode & F4L_ICP_INIT_ROUND_F32) ? 1 : 0;
    fx.normals_f64 = (mode & F4L_ICP_NORMALS_F64) ? 1 : 0;
    fx.p2pl_open3d = (mode & F4L_ICP_P2PL_OPEN3D) ? 1 : 0;
    mode &= ~(F4L_ICP_INIT_ROUND_F32 | F4L_ICP_NORMALS_F64 | F4L_ICP_P2PL_OPEN3D);
    return f4l::icp_launch_host(src, src_off, tgt, tgt_off, P, nullptr, tgt_normals, max_corr_dist, max_iter, rel_fitness,
                                rel_rmse, mode, fixed_iters, search_precision, max_src_patch_host, max_tgt_patch_host,
                                n_src_host, T_out, fitness_out, rmse_out, iters_out, corr_out, fx, stream);
}

static int f4l::icp_launch_host(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                                int64_t P, const double *init_T, const float *tgt_normals, double max_corr_dist,
                                int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                                int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host,
                                int64_t n_src_host, double *T_out, double *fitness_out, double *rmse_out,
                                int32_t *iters_out, int32_t *corr_out, const IcpFusedExtra &fx, void *stream) {
    using namespace f4l;
    if (P < 0 || !src_off || !tgt_off || !T_out || max_iter < 0 || max_src_patch_host < 0 || max_tgt_patch_host < 0 ||
        n_src_host < 0)
        return F4L_EINVAL;
    if (mode != F4L_ICP_POINT2POINT && mode != F4L_ICP_POINT2PLANE && mode != F4L_ICP_GENERALIZED) return F4L_EINVAL;
    if (search_precision != F4L_SEARCH_F32 && search_precision != F4L_SEARCH_F64) return F4L_EINVAL;
    if (mode != F4L_ICP_POINT2POINT && max_tgt_patch_host > 0 && !tgt_normals) return F4L_EINVAL;
    if (mode == F4L_ICP_GENERALIZED && ((max_src_patch_host > 0 && !fx.src_normals) || !fx.normals_f64 || !(fx.gicp_eps >= 0.0)))
        return F4L_EINVAL;
    if ((max_src_patch_host > 0 && !src) || (max_tgt_patch_host > 0 && !tgt)) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_src_patch_host > 0x3fffffffLL || max_tgt_patch_host > 0x3fffffffLL)
        return F4L_EUNSUPPORTED;
    // Point-to-plane always measures in double: its 6 x 6 system is ill-conditioned on near-planar or half-matched patches, and the
    // 1e-7 of a float32 position, harmless to the Umeyama sums, was seen to throw such a patch out of reach of every target
    // (fitness 0 after 20 passes where the float64 search follows the oracle to 1e-8 m: tools/gpu/fuzz_icp.py 1 2002788 f32).
    const bool f64 = search_precision == F4L_SEARCH_F64 || mode != F4L_ICP_POINT2POINT;
    static_assert(sizeof(GridPt<double>) == sizeof(GridPt<float>), "grid records are 16 B in both modes");
    IcpArgs a;
    a.src = src; a.src_off = src_off; a.tgt = tgt; a.tgt_off = tgt_off; a.P = P;
    a.init_T = init_T; a.tgt_normals = tgt_normals;
    a.corr_src = fx.corr_src; a.corr_ref = fx.corr_ref; a.corr_w = fx.corr_w; a.corr_off = fx.corr_off;
    a.kabsch_w_thresh = fx.w_thresh; a.kabsch_eps = fx.eps; a.rows_out = fx.rows_out;
    a.rows_src = fx.rows_src; a.rows_off = fx.rows_off; a.min_corr = fx.min_corr; a.init_round_f32 = fx.init_round_f32;
    a.normals_f64 = fx.normals_f64;
    a.p2pl_open3d = fx.p2pl_open3d;
    a.src_normals = fx.src_normals; a.gicp_eps = fx.gicp_eps;
    a.r = max_corr_dist > 0.0 ? max_corr_dist : 0.0;
    a.r2 = a.r * a.r;
    a.max_iter = max_iter; a.rel_fitness = rel_fitness; a.rel_rmse = rel_rmse; a.fixed_iters = fixed_iters;
    { const char *dbg = getenv("F4L_ICP_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
    a.subdiv = 8; a.mu_frac = 0.125; a.dens = 2.f;
    a.xsub = 8;  // (C3 15.4 ms with 8, 15.7 with 4, 16.8 with 2, 20.1 with cubic cells; C4 18.3 / 18.3 / 19.0 / 19.3)
    { const char *e = getenv("F4L_ICP_XSUB"); if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8)) a.xsub = atoi(e); }
    { const char *e = getenv("F4L_ICP_DENS"); if (e && atof(e) > 0.0) a.dens = (float)atof(e); }
    { const char *e = getenv("F4L_ICP_SUBDIV"); if (e && atoi(e) >= 1) a.subdiv = atoi(e); }
    { const char *e = getenv("F4L_ICP_MU"); if (e && atof(e) > 0.0) a.mu_frac = atof(e); }
    // (an eighth of a cell: C3 20.3 ms against 22.3 with a quarter -- a search scans everything within the previous
    //  correspondence's distance plus this margin, and on a fine grid the runner-up limits the certificate long before it)
    a.mu_cell = 0.125;
    // (grids finer than the radius: a sixteenth -- C3 14.8 ms against 15.3 with an eighth and 17.1 with a quarter, round 4)
    a.mu_cell_fine = 0.0625;
    { const char *e = getenv("F4L_ICP_MU_CELL"); if (e && atof(e) > 0.0) a.mu_cell = a.mu_cell_fine = atof(e); }
    a.T_out = T_out; a.fitness_out = fitness_out; a.rmse_out = rmse_out; a.iters_out = iters_out; a.corr_out = corr_out;

    // throughput regime: six and more rounds of workgroups (1024 slots of four waves on the chip), where patches per second
    // count and not one patch's latency: measured crossover between 2025 patches (C2: 0.68 ms against 0.80 ms in the
    // throughput shape) and 8100 (2.08 against 2.30 ms; 32 400: 6.6 against 8.5 ms).  Batches with patches far larger than
    // the rest (the size classes below; C3, whose patches are also much denser than the radius) stay with four waves:
    // 27.1 against 30.0 ms there.
    bool throughput = P >= 6144 && icp_plan(max_src_patch_host, max_tgt_patch_host, f64, mode, false).lds <= 48 * 1024;
    if (const char *e = getenv("F4L_ICP_THROUGHPUT")) throughput = atoi(e) != 0;
    // Several small patches per wave (icp_rows.h) pay where the launch is bound by issue slots, not by one wave's latency, and
    // where the kernel does not spill: the float32 search from ~64 k patches on (10 M-point tile cut into 167 k supervoxel
    // patches: 6.05 ms against 7.7 ms; float64: 9.0 against 8.8 ms -- 432 B of scratch per lane; the 1 M-point tile, 16.7 k
    // patches, is slower with it in both modes: 1.57 against 1.22 ms).  F4L_ICP_ROWS = 1 / 0 forces it on / off.
    bool rows_ok = !f64 && P >= 65536;
    if (const char *e = getenv("F4L_ICP_ROWS")) rows_ok = atoi(e) != 0;
    const IcpPlan pl = icp_plan(max_src_patch_host, max_tgt_patch_host, f64, mode, throughput, rows_ok);
    const int nw = pl.nw, tgt_cap = pl.tgt_cap, cell_cap = pl.cell_cap, cert_cap = pl.cert_cap, src_cap = pl.src_cap;
    const size_t lds = pl.lds;
    a.tgt_cap = tgt_cap; a.cert_cap = cert_cap; a.cell_cap = cell_cap; a.src_cap = src_cap; a.pp_cap = pl.pp_cap;
    a.list = nullptr; a.list_cnt = nullptr;
    a.prof = nullptr;
#ifdef F4L_ICP_PROF
    if (getenv("F4L_ICP_PROF")) {
        unsigned long long *dp = nullptr, hp[32];
        const size_t prof_bytes = sizeof(hp) + (size_t)P * 16;
        F4L_HIP_CHECK(hipMalloc(&dp, prof_bytes));
        F4L_HIP_CHECK(hipMemset(dp, 0, prof_bytes));
        a.prof = dp;
        a.prof_max_n = getenv("F4L_ICP_PROF_MAXN") ? atoi(getenv("F4L_ICP_PROF_MAXN")) : 0x7fffffff;
        int rc = f64 ? launch_icp<double>(a, mode, nw, lds, (hipStream_t)stream, pl.wide) : launch_icp<float>(a, mode, nw, lds, (hipStream_t)stream, pl.wide);
        F4L_HIP_CHECK(hipDeviceSynchronize());
        F4L_HIP_CHECK(hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost));
        if (getenv("F4L_ICP_PROF_WG")) {  // per-workgroup start/end (100 MHz ticks) -> schedule statistics
            unsigned long long *wg = (unsigned long long *)malloc((size_t)P * 16);
            F4L_HIP_CHECK(hipMemcpy(wg, dp + 32, (size_t)P * 16, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ULL, t1 = 0, dmin = ~0ULL, dmax = 0;
            double dsum = 0;
            for (int64_t i = 0; i < P; ++i) {
                const unsigned long long b = wg[2 * i], e = wg[2 * i + 1], d = e - b;
                t0 = b < t0 ? b : t0; t1 = e > t1 ? e : t1; dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; dsum += (double)d;
            }
            fprintf(stderr, "[icp prof wg] span %.1f us | per WG: min %.1f mean %.1f max %.1f us | mean concurrency %.0f WGs\n",
                    (t1 - t0) * 0.01, dmin * 0.01, dsum / P * 0.01, dmax * 0.01, dsum / (double)(t1 - t0));
            const int NB = 20;
            for (int b = 0; b < NB; ++b) {  // running workgroups at 20 instants
                const unsigned long long t = t0 + (t1 - t0) * (unsigned long long)(2 * b + 1) / (2 * NB);
                int live = 0;
                for (int64_t i = 0; i < P; ++i) live += (wg[2 * i] <= t && t < wg[2 * i + 1]) ? 1 : 0;
                fprintf(stderr, "%d ", live);
            }
            fprintf(stderr, "\n");
            if (FILE *f = fopen(getenv("F4L_ICP_PROF_WG"), "wb")) { fwrite(wg, 16, (size_t)P, f); fclose(f); }
            free(wg);
        }
        hipFree(dp);
        const double Pw = (double)(hp[26] ? hp[26] : 1);
        fprintf(stderr, "[icp prof] P=%lld nw=%d lds=%zu tgt_cap=%d cert_cap=%d src_cap=%d cell_cap=%d | per-WG mean cycles: total %.0f build %.0f phase1 %.0f search %.0f reduce %.0f solve %.0f barrier %.0f dpp %.0f clock %.3f GHz | per query: steps %.2f rows %.2f rows_taken %.2f wave-steps/batch %.2f | per WG-pass: searched %.1f of %.1f, wave-batches %.2f\n",
                (long long)Pw, nw, lds, tgt_cap, cert_cap, src_cap, cell_cap, hp[0] / (double)Pw, hp[1] / (double)Pw, hp[15] / (double)Pw, hp[2] / (double)Pw,
                hp[3] / (double)Pw, hp[4] / (double)Pw, hp[5] / (double)Pw, hp[12] / (double)Pw, hp[0] / (double)(hp[14] ? hp[14] : 1) * 0.1, hp[6] / (double)(hp[9] ? hp[9] : 1),
                hp[7] / (double)(hp[9] ? hp[9] : 1), hp[8] / (double)(hp[9] ? hp[9] : 1), hp[10] / (double)(hp[11] ? hp[11] : 1), hp[9] / (double)Pw / (max_iter + 1), (double)max_src_patch_host, hp[11] / (double)Pw / (max_iter + 1));
        fprintf(stderr, "[icp prof wide] queries %llu wave-calls %llu mean W %.2f | scan rounds %llu, stencil rows per round %.2f, rows kept per lane-round %.2f, steps per lane-round %.2f\n",
                hp[23], hp[24], hp[25] / (double)(hp[24] ? hp[24] : 1), hp[13], hp[22] / (double)(hp[13] ? hp[13] : 1),
                hp[21] / (double)(hp[13] ? hp[13] : 1) / 64.0, hp[20] / (double)(hp[13] ? hp[13] : 1) / 64.0);
        fprintf(stderr, "[icp prof grid_nn] cycles per wave-call: narrow %.0f (%llu calls), wide %.0f; wide stencil rows visited per call %.1f; calls with W >= 8: %llu\n",
                hp[27] / (double)((hp[11] > hp[24] ? hp[11] - hp[24] : 1)), (unsigned long long)(hp[11] - hp[24]), hp[28] / (double)(hp[24] ? hp[24] : 1),
                hp[29] / (double)(hp[24] ? hp[24] : 1), hp[30]);
        return rc;
    }
#endif
    // One LDS size per launch: when the largest patch would leave a CU with two workgroups or fewer, patches are
    // binned by size on the device and every class is launched with its own plan, largest first.  The host does not
    // know the class sizes (no synchronisation here): each launch has P workgroups, those past its class count return.
    const int64_t big = max_src_patch_host > max_tgt_patch_host ? max_src_patch_host : max_tgt_patch_host;
    ClassBounds cb;
    cb.n = 0;
    const bool lds_classes = lds > 48 * 1024;
    // ... and when the patches are of uneven size (mean below 3/4 of the largest, known only if the caller passed the
    // point count) patches of one or two wavefronts get workgroups of just those: 1.75x on a supervoxel partition
    // (median 58 points, largest 153).  Evenly sized patches (C2: mean 494, largest 574) stay one launch.
    const bool wave_classes = nw > 1 && P >= 512 &&
                              (getenv("F4L_ICP_SMALLCLASSES") || (n_src_host > 0 && 4 * n_src_host <= 3 * P * max_src_patch_host));
    // ... and in the throughput regime, when a few patches are much larger than the rest (border patches that collect what
    // moved out of the tile: 902 targets against a mean of 500 at C4), the bulk gets a class of its own whose LDS is sized
    // for IT: one more workgroup per CU for 99 % of the patches.
    int64_t bulk = 0;
    if (throughput && n_src_host > 0 && P >= 4096 && !getenv("F4L_ICP_NOSPLIT")) {
        const int64_t mean = n_src_host / P;
        const int64_t b = ((5 * mean / 4) + 63) / 64 * 64;
        if (b > 128 && 13 * mean <= 10 * big && b < big) bulk = b;
    }
    if ((lds_classes || wave_classes || bulk) && P >= 64 && !getenv("F4L_ICP_NOCLASSES"))
    {
        // ratio between class bounds: 4 (256, 1024, 4096), 2, or 1 = ~sqrt(2) (256, 384, 512, 768, ...): every class pays
        // for its own largest patch only -- C3 27.3 / 24.3 / 22.3 ms
        int step = 1;
        if (const char *e = getenv("F4L_ICP_CLASS_STEP")) step = atoi(e);
        if (wave_classes)  // patches that fit one or two wavefronts get workgroups of just those
            for (int64_t b = 64; b <= 128 && 3 * b <= 2 * big; b *= 2) cb.bound[cb.n++] = (int)b;
        for (int64_t b = 256; lds_classes && b <= 4096 && 3 * b <= 2 * big && cb.n < ICP_MAX_CLASSES - 2;) {
            cb.bound[cb.n++] = (int)b;
            b = step == 4 ? b * 4 : (step == 2 ? b * 2 : ((cb.n & 1) ? (b * 3) / 2 : (b * 4) / 3));
        }
        if (bulk && cb.n < ICP_MAX_CLASSES - 2) {  // insert in ascending order, unless a bound that close exists already
            int at = 0;
            while (at < cb.n && cb.bound[at] < bulk) ++at;
            const bool near = (at < cb.n && cb.bound[at] <= bulk + bulk / 4) || (at > 0 && cb.bound[at - 1] >= bulk - bulk / 4);
            if (!near) {
                for (int i = cb.n; i > at; --i) cb.bound[i] = cb.bound[i - 1];
                cb.bound[at] = (int)bulk;
                ++cb.n;
            }
        }
    }
    if (cb.n == 0 && !pl.rows_lp)
        return f64 ? launch_icp<double>(a, mode, nw, lds, (hipStream_t)stream, pl.wide)
                   : launch_icp<float>(a, mode, nw, lds, (hipStream_t)stream, pl.wide);
    cb.bound[cb.n++] = (int)(big > 0x7fffffff ? 0x7fffffff : big);
    // icp_rows_kernel holds at most ROWS_PPL * LP sources and targets per patch and, unlike icp_kernel, has no path for a patch
    // beyond that: it must only ever see patches icp_bin_patches has MEASURED (the caller's max_*_patch may be understated --
    // ADVICE r4).  So a launch that would give the last class to the rows kernel bounds that class by the kernel's capacity and
    // adds one class behind it, planned for icp_kernel, which takes whatever is larger than the caller said (normally nothing:
    // its workgroups return at once).
    int overflow_class = -1;
    if (icp_plan(big, big, f64, mode, throughput, rows_ok).rows_lp) {
        cb.bound[cb.n - 1] = ROWS_PPL * icp_plan(big, big, f64, mode, throughput, rows_ok).rows_lp;
        overflow_class = cb.n;
        cb.bound[cb.n++] = 0x7fffffff;
    }
    hipStream_t st = (hipStream_t)stream;
    int *buf = nullptr;
    const size_t buf_bytes = ((size_t)cb.n * (size_t)P + (size_t)cb.n) * sizeof(int);
    // One exit for every outcome (`fail` records the HIP error and jumps there): whatever was launched on a helper stream is
    // joined back to the caller's stream, the events are destroyed and the binning buffer is released before returning, so
    // that an error return never leaves side-stream kernels reading the caller's tensors or leaks the temporaries.
    int rc = F4L_OK;
    hipEvent_t forked = nullptr, joined[ICP_MAX_CLASSES] = {};
    hipStream_t used[ICP_MAX_CLASSES] = {};
    auto fail = [&](hipError_t e) {
        if (e == hipSuccess) return false;
        f4l_tls_hip_error = (int)e;
        rc = F4L_EHIP;
        return true;
    };
    do {
        if (fail(hipMallocAsync((void **)&buf, buf_bytes, st))) { buf = nullptr; break; }
        int *cnt = buf + (size_t)cb.n * (size_t)P;
        if (fail(hipMemsetAsync(cnt, 0, (size_t)cb.n * sizeof(int), st))) break;
        hipLaunchKernelGGL(icp_bin_patches, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, src_off, tgt_off, (int)P, cb, cnt, buf);
        if (fail(hipGetLastError())) break;
        // The classes are independent and none of them fills the machine to its end (a class of a few large patches is one
        // long chain on a few CUs): they run side by side, the largest on the caller's stream, the others on streams of
        // this library's own that fork from it after the binning and join it again (events; no host synchronisation).
        const bool side = !getenv("F4L_ICP_SERIAL_CLASSES");
        if (side) {
            if (fail(hipEventCreateWithFlags(&forked, hipEventDisableTiming))) { forked = nullptr; break; }
            if (fail(hipEventRecord(forked, st))) break;
        }
        for (int k = cb.n - 1; k >= 0 && rc == F4L_OK; --k) {
            const int64_t ms = max_src_patch_host < cb.bound[k] ? max_src_patch_host : cb.bound[k];
            const int64_t mt = max_tgt_patch_host < cb.bound[k] ? max_tgt_patch_host : cb.bound[k];
            const IcpPlan pk = icp_plan(ms, mt, f64, mode, throughput, rows_ok && k != overflow_class);
            IcpArgs ak = a;
            ak.tgt_cap = pk.tgt_cap; ak.cert_cap = pk.cert_cap; ak.cell_cap = pk.cell_cap; ak.src_cap = pk.src_cap; ak.pp_cap = pk.pp_cap;
            ak.list = buf + (size_t)k * (size_t)P; ak.list_cnt = cnt + k;
            hipStream_t sk = st;
            if (side && k != cb.n - 1) {
                sk = class_stream(cb.n - 2 - k);
                if (!sk) { rc = F4L_EHIP; break; }
                if (fail(hipStreamWaitEvent(sk, forked, 0))) break;
                used[k] = sk;  // from here on the helper stream may hold work that reads the caller's buffers
            }
            if (pk.rows_lp) rc = f64 ? launch_rows<double>(ak, pk.rows_lp, sk) : launch_rows<float>(ak, pk.rows_lp, sk);
            else rc = f64 ? launch_icp<double>(ak, mode, pk.nw, pk.lds, sk, pk.wide) : launch_icp<float>(ak, mode, pk.nw, pk.lds, sk, pk.wide);
        }
    } while (false);
    // join every helper stream that was handed work (also after an error), then release
    for (int k = 0; k < ICP_MAX_CLASSES; ++k) {
        if (!used[k]) continue;
        if (hipEventCreateWithFlags(&joined[k], hipEventDisableTiming) != hipSuccess) { joined[k] = nullptr; (void)hipStreamSynchronize(used[k]); continue; }
        if (hipEventRecord(joined[k], used[k]) != hipSuccess || hipStreamWaitEvent(st, joined[k], 0) != hipSuccess) {
            if (rc == F4L_OK) { f4l_tls_hip_error = (int)hipGetLastError(); rc = F4L_EHIP; }
            (void)hipStreamSynchronize(used[k]);  // last resort: the caller's buffers must outlive the helper's kernels
        }
    }
    // (destroying an event whose work is still in flight only defers the release)
    if (forked) (void)hipEventDestroy(forked);
    for (int k = 0; k < ICP_MAX_CLASSES; ++k)
        if (joined[k]) (void)hipEventDestroy(joined[k]);
    if (buf && hipFreeAsync(buf, st) != hipSuccess && rc == F4L_OK) { f4l_tls_hip_error = (int)hipGetLastError(); rc = F4L_EHIP; }
    return rc;
}
#endif  // F4L_ICP_BULK_TU
