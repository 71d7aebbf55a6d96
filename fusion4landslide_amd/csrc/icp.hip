// icp.hip -- batched per-patch ICP (point-to-point and point-to-plane) for gfx950.
//
// Replaces, for P patch pairs in ONE launch and with zero host round trips, the loop body
//   src/coarse_to_fine_matching_base.py:3353-3367  ->  utils/o3d_tools.py:12-71 `icp_registration`
//   ->  Open3D 0.19 registration_icp(point2point | point2plane, criteria(1e-6, 1e-6, 30))
// which in the reference copies every patch GPU->CPU, builds a KD-tree, iterates on the CPU and copies
// the 4x4 back (two device crossings per patch).
//
// Mapping to CDNA4
//   * one 256-thread workgroup (4 waves) per patch pair;
//   * the target patch is staged ONCE into LDS as float4 (x,y,z relative to a per-patch origin) and stays
//     resident for every iteration; all 64 lanes of a wave read the same candidate -> LDS broadcast reads,
//     conflict free (ds_read_b128, one per candidate per wave);
//   * each lane owns one or two source points per pass (registers), brute-force nearest neighbour inside the
//     patch: 3 sub + 1 mul + 2 fma + compare/select per pair, no MFMA (the contraction is 3x3);
//   * per pass the correspondence sums (17 doubles for Umeyama, 29 for the 6x6 point-to-plane system) are
//     reduced with wavefront __shfl butterflies, then across the 4 waves through LDS; wave 0 solves the 3x3
//     Jacobi SVD / 6x6 system in double and broadcasts the new transform through LDS;
//   * convergence test, iteration count, fitness and rmse are evaluated on the device.
// Source points are re-read from global memory each pass (12 B/pt/iter, L2 resident after the first pass);
// with the target share this is the 24 B/pt/iter algorithmic traffic of SURVEY.md 8(d).
//
// Numerics: coordinates are taken relative to a per-patch origin (first target point) so that float32
// distance arithmetic works at ~1 m magnitudes even for georeferenced clouds; the running transform and all
// sums are double.
#include <stdlib.h>

#include "f4l_device.h"

namespace f4l {

constexpr int ICP_NW = 4;             // waves per workgroup
constexpr int ICP_NT = ICP_NW * 64;   // threads per workgroup
constexpr int ICP_LDS_TGT_MAX = 8192; // target points kept in LDS at most (128 KiB of the 160 KiB)

struct IcpArgs {
    const float *src;
    const int64_t *src_off;
    const float *tgt;
    const int64_t *tgt_off;
    int64_t P;
    const double *init_T;
    const float *tgt_normals;
    double r2;
    int max_iter;
    double rel_fitness, rel_rmse;
    int fixed_iters;
    int lds_cap;  // number of float4 target slots in dynamic LDS
    double *T_out, *fitness_out, *rmse_out;
    int32_t *iters_out, *corr_out;
    int debug;  // F4L_ICP_DEBUG env: 1 = skip the solve, 2 = skip the search (timing experiments only)
};

// Target point as staged in LDS: float4 (16 B, one ds_read_b128) for the float32 search, 3 doubles for the
// float64 ("reference arithmetic") search.
template <typename F> struct TgtPt;
template <> struct TgtPt<float> { float x, y, z, w; };
template <> struct TgtPt<double> { double x, y, z; };

template <typename F> __device__ __forceinline__ F f_inf();
template <> __device__ __forceinline__ float f_inf<float>() { return __builtin_inff(); }
template <> __device__ __forceinline__ double f_inf<double>() { return __builtin_inf(); }

// Nearest target of up to two query points; targets in LDS (origin-relative).
template <int SPT, typename F>
__device__ __forceinline__ void nn_lds(const TgtPt<F> *__restrict__ tl, int nt, const F (&px)[2], const F (&py)[2],
                                       const F (&pz)[2], F (&best)[2], int (&bj)[2]) {
#pragma unroll 8
    for (int j = 0; j < nt; ++j) {
        const TgtPt<F> q = tl[j];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const F dx = px[s] - q.x, dy = py[s] - q.y, dz = pz[s] - q.z;
            const F d = dx * dx + dy * dy + dz * dz;
            if (d < best[s]) { best[s] = d; bj[s] = j; }
        }
    }
}

// Same with targets in global memory (patches larger than the LDS budget): wave-uniform addresses.
template <int SPT, typename F>
__device__ __forceinline__ void nn_global(const float *__restrict__ tg, int nt, float ox, float oy, float oz,
                                          const F (&px)[2], const F (&py)[2], const F (&pz)[2], F (&best)[2],
                                          int (&bj)[2]) {
#pragma unroll 4
    for (int j = 0; j < nt; ++j) {
        const F qx = (F)tg[3 * j] - (F)ox, qy = (F)tg[3 * j + 1] - (F)oy, qz = (F)tg[3 * j + 2] - (F)oz;
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const F dx = px[s] - qx, dy = py[s] - qy, dz = pz[s] - qz;
            const F d = dx * dx + dy * dy + dz * dz;
            if (d < best[s]) { best[s] = d; bj[s] = j; }
        }
    }
}

// Solve the 6x6 system M[:, :6] x = M[:, 6] in place by Gaussian elimination with partial pivoting.
// All indices are compile-time after unrolling, so M lives in registers.  Returns false when singular.
__device__ __forceinline__ bool solve6(double (&M)[6][7], double (&x)[6]) {
    bool ok = true;
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
        if (piv == 0.0 || !isfinite(piv)) ok = false;
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

// LDS layout (dynamic): [ scratch doubles | state doubles | float4 targets ]
//   scratch : ICP_NW * NV partial sums
//   state   : Rc[9], tc[3], flag      (flag: 0 continue, 1 finished)
template <int MODE, int SPT, typename F>
__global__ __launch_bounds__(ICP_NT, 4) void icp_kernel(IcpArgs a) {
    constexpr int NV = (MODE == F4L_ICP_POINT2POINT) ? 17 : 29;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);
    double *state = scratch + ICP_NW * 32;  // 32 >= NV keeps the float4 region 16-byte aligned
    TgtPt<F> *tl = reinterpret_cast<TgtPt<F> *>(state + 16);

    const int64_t p = blockIdx.x;
    if (p >= a.P) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = a.src_off[p], t0 = a.tgt_off[p];
    const int ns = (int)(a.src_off[p + 1] - s0), nt = (int)(a.tgt_off[p + 1] - t0);
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    const bool in_lds = nt <= a.lds_cap;

    // per-patch origin: first target point (else first source point, else 0)
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    else if (ns > 0) { ox = sg[0]; oy = sg[1]; oz = sg[2]; }

    if (in_lds) {
        for (int j = tid; j < nt; j += ICP_NT) {
            TgtPt<F> q;
            q.x = (F)tg[3 * j] - (F)ox; q.y = (F)tg[3 * j + 1] - (F)oy; q.z = (F)tg[3 * j + 2] - (F)oz;
            tl[j] = q;
        }
    }

    // running transform in origin-relative coordinates, p' = Rc s' + tc, lives in LDS `state`
    // (state[0..8] = Rc, [9..11] = tc, [12] = done flag, [13] = fitness, [14] = rmse, [15] = iterations)
    if (tid == 0) {
        if (a.init_T) {
            const double *T = a.init_T + 16 * p;
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
    }
    __syncthreads();

    const bool active = ns > 0 && a.r2 > 0.0;
    const F r2 = (F)a.r2;  // o3d returns the init untouched when max_corr_dist <= 0
    const int n_pass = active ? a.max_iter + 1 : 0;

    for (int pass = 0; pass < n_pass; ++pass) {
        const F R0 = (F)state[0], R1 = (F)state[1], R2 = (F)state[2], R3 = (F)state[3], R4 = (F)state[4],
                R5 = (F)state[5], R6 = (F)state[6], R7 = (F)state[7], R8 = (F)state[8];
        const F t0f = (F)state[9], t1f = (F)state[10], t2f = (F)state[11];
        double acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.0;

        for (int base = 0; base < ns; base += ICP_NT * SPT) {
            F px[2], py[2], pz[2], best[2];
            int bj[2], si[2];
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                si[s] = base + s * ICP_NT + tid;
                const int ii = si[s] < ns ? si[s] : ns - 1;  // clamp: inactive lanes recompute the last point
                const F x = (F)sg[3 * ii] - (F)ox, y = (F)sg[3 * ii + 1] - (F)oy, z = (F)sg[3 * ii + 2] - (F)oz;
                px[s] = R0 * x + R1 * y + R2 * z + t0f;
                py[s] = R3 * x + R4 * y + R5 * z + t1f;
                pz[s] = R6 * x + R7 * y + R8 * z + t2f;
                best[s] = f_inf<F>();
                bj[s] = -1;
            }
            if (a.debug & 2) { bj[0] = 0; bj[1] = 0; best[0] = (F)1e-4; best[1] = (F)1e-4; }
            else if (in_lds) nn_lds<SPT, F>(tl, nt, px, py, pz, best, bj);
            else nn_global<SPT, F>(tg, nt, ox, oy, oz, px, py, pz, best, bj);
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                const bool valid = si[s] < ns;
                const bool hit = valid && bj[s] >= 0 && best[s] < r2;  // SearchHybrid: d2 < r^2
                if (a.corr_out && valid) a.corr_out[s0 + si[s]] = hit ? bj[s] : -1;
                if (hit) {
                    F qx, qy, qz;
                    if (in_lds) { const TgtPt<F> q = tl[bj[s]]; qx = q.x; qy = q.y; qz = q.z; }
                    else { qx = (F)tg[3 * bj[s]] - (F)ox; qy = (F)tg[3 * bj[s] + 1] - (F)oy; qz = (F)tg[3 * bj[s] + 2] - (F)oz; }
                    const double dpx = px[s], dpy = py[s], dpz = pz[s], dqx = qx, dqy = qy, dqz = qz;
                    acc[0] += 1.0;
                    acc[1] += (double)best[s];
                    if (MODE == F4L_ICP_POINT2POINT) {
                        acc[2] += dpx; acc[3] += dpy; acc[4] += dpz;
                        acc[5] += dqx; acc[6] += dqy; acc[7] += dqz;
                        acc[8] += dqx * dpx; acc[9] += dqx * dpy; acc[10] += dqx * dpz;
                        acc[11] += dqy * dpx; acc[12] += dqy * dpy; acc[13] += dqy * dpz;
                        acc[14] += dqz * dpx; acc[15] += dqz * dpy; acc[16] += dqz * dpz;
                    } else {
                        const float *nn = a.tgt_normals + 3 * (t0 + bj[s]);
                        const double nx = nn[0], ny = nn[1], nz = nn[2];
                        const double r = (dpx - dqx) * nx + (dpy - dqy) * ny + (dpz - dqz) * nz;
                        double J[6];
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
                }
            }
        }

        // wave butterflies, then the 4 partials through LDS; wave 0 solves
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) scratch[wave * 32 + i] = acc[i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                acc[i] = scratch[i] + scratch[32 + i] + scratch[64 + i] + scratch[96 + i];
            double Rc[9], tc[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rc[i] = state[i];
            tc[0] = state[9]; tc[1] = state[10]; tc[2] = state[11];
            const double fitness = state[13], rmse = state[14];
            int iters = (int)state[15];
            const double m = acc[0];
            const double fit_new = m > 0.0 ? m / (double)ns : 0.0;
            const double rmse_new = m > 0.0 ? sqrt(acc[1] / m) : 0.0;
            bool done = false;
            if (pass > 0) {
                iters = pass;
                if (!a.fixed_iters && fabs(fitness - fit_new) < a.rel_fitness && fabs(rmse - rmse_new) < a.rel_rmse)
                    done = true;
            }
            if (pass == a.max_iter) done = true;
            if (!done && m > 0.0 && !(a.debug & 1)) {
                double Ru[9], tu[3];
                bool have = true;
                if (MODE == F4L_ICP_POINT2POINT) {
                    // Eigen::umeyama without scaling
                    const double im = 1.0 / m;
                    const double mp0 = acc[2] * im, mp1 = acc[3] * im, mp2 = acc[4] * im;
                    const double mq0 = acc[5] * im, mq1 = acc[6] * im, mq2 = acc[7] * im;
                    double sg9[9];
                    sg9[0] = acc[8] * im - mq0 * mp0; sg9[1] = acc[9] * im - mq0 * mp1; sg9[2] = acc[10] * im - mq0 * mp2;
                    sg9[3] = acc[11] * im - mq1 * mp0; sg9[4] = acc[12] * im - mq1 * mp1; sg9[5] = acc[13] * im - mq1 * mp2;
                    sg9[6] = acc[14] * im - mq2 * mp0; sg9[7] = acc[15] * im - mq2 * mp1; sg9[8] = acc[16] * im - mq2 * mp2;
                    double U[9], S[3], V[9];
                    svd3(sg9, U, S, V);
                    const double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
                    mul_diag_bt(U, sgn, V, Ru);
                    tu[0] = mq0 - (Ru[0] * mp0 + Ru[1] * mp1 + Ru[2] * mp2);
                    tu[1] = mq1 - (Ru[3] * mp0 + Ru[4] * mp1 + Ru[5] * mp2);
                    tu[2] = mq2 - (Ru[6] * mp0 + Ru[7] * mp1 + Ru[8] * mp2);
                } else {
                    double M[6][7], x[6];
                    int k = 2;
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int v = u; v < 6; ++v) { M[u][v] = acc[k]; M[v][u] = acc[k]; ++k; }
#pragma unroll
                    for (int u = 0; u < 6; ++u) M[u][6] = -acc[23 + u];
                    have = solve6(M, x);
                    if (have) {
                        // o3d TransformVector6dToMatrix4d: Rz(x2) Ry(x1) Rx(x0), translation x[3:6]
                        const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]),
                                     sgm = sin(x[2]);
                        Ru[0] = cg * cb; Ru[1] = cg * sb * sa - sgm * ca; Ru[2] = cg * sb * ca + sgm * sa;
                        Ru[3] = sgm * cb; Ru[4] = sgm * sb * sa + cg * ca; Ru[5] = sgm * sb * ca - cg * sa;
                        Ru[6] = -sb; Ru[7] = cb * sa; Ru[8] = cb * ca;
                        // the system was built about the patch origin o: t' = t + x cross o.  Undo that, then express
                        // the reference's update [Rot(x) | t] (a rotation about the GLOBAL origin) in origin-relative
                        // coordinates: tu = Rot(x) o + t - o.
                        const double o0 = ox, o1 = oy, o2 = oz;
                        const double tg0 = x[3] - (x[1] * o2 - x[2] * o1);
                        const double tg1 = x[4] - (x[2] * o0 - x[0] * o2);
                        const double tg2 = x[5] - (x[0] * o1 - x[1] * o0);
                        tu[0] = Ru[0] * o0 + Ru[1] * o1 + Ru[2] * o2 + tg0 - o0;
                        tu[1] = Ru[3] * o0 + Ru[4] * o1 + Ru[5] * o2 + tg1 - o1;
                        tu[2] = Ru[6] * o0 + Ru[7] * o1 + Ru[8] * o2 + tg2 - o2;
                    }
                }
                if (have) {  // T <- update * T
                    double Rn[9], tn[3];
                    mul3(Ru, Rc, Rn);
                    tn[0] = Ru[0] * tc[0] + Ru[1] * tc[1] + Ru[2] * tc[2] + tu[0];
                    tn[1] = Ru[3] * tc[0] + Ru[4] * tc[1] + Ru[5] * tc[2] + tu[1];
                    tn[2] = Ru[6] * tc[0] + Ru[7] * tc[1] + Ru[8] * tc[2] + tu[2];
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rn[i];
                    tc[0] = tn[0]; tc[1] = tn[1]; tc[2] = tn[2];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) state[i] = Rc[i];
                state[9] = tc[0]; state[10] = tc[1]; state[11] = tc[2];
                state[12] = done ? 1.0 : 0.0;
                state[13] = fit_new; state[14] = rmse_new; state[15] = (double)iters;
            }
        }
        __syncthreads();
        const bool finished = state[12] != 0.0;
        if (finished) break;  // uniform across the workgroup
    }

    if (tid == 0) {
        // back to global coordinates: t = tc - Rc o + o
        const double o0 = ox, o1 = oy, o2 = oz;
        double Rc[9], tc[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rc[i] = state[i];
        tc[0] = state[9]; tc[1] = state[10]; tc[2] = state[11];
        const double fitness = state[13], rmse = state[14];
        const int iters = (int)state[15];
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
        for (int i = tid; i < ns; i += ICP_NT) a.corr_out[s0 + i] = -1;
}

}  // namespace f4l

namespace f4l {
template <typename F>
static int launch_icp(const IcpArgs &a, int mode, bool two, size_t lds, hipStream_t st) {
    dim3 grid((unsigned)a.P), block(ICP_NT);
    const void *fn;
    if (mode == F4L_ICP_POINT2POINT) fn = two ? (const void *)icp_kernel<0, 2, F> : (const void *)icp_kernel<0, 1, F>;
    else fn = two ? (const void *)icp_kernel<1, 2, F> : (const void *)icp_kernel<1, 1, F>;
    if (lds > 64 * 1024)  // opt in to > 64 KiB of dynamic LDS
        F4L_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (mode == F4L_ICP_POINT2POINT) {
        if (two) hipLaunchKernelGGL((icp_kernel<0, 2, F>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((icp_kernel<0, 1, F>), grid, block, lds, st, a);
    } else {
        if (two) hipLaunchKernelGGL((icp_kernel<1, 2, F>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((icp_kernel<1, 1, F>), grid, block, lds, st, a);
    }
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
}  // namespace f4l

extern "C" int f4l_piecewise_icp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                                 int64_t P, const double *init_T, const float *tgt_normals, double max_corr_dist,
                                 int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                                 int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host,
                                 double *T_out, double *fitness_out, double *rmse_out, int32_t *iters_out,
                                 int32_t *corr_out, void *stream) {
    using namespace f4l;
    if (P < 0 || !src_off || !tgt_off || !T_out || max_iter < 0 || max_src_patch_host < 0 || max_tgt_patch_host < 0)
        return F4L_EINVAL;
    if (mode != F4L_ICP_POINT2POINT && mode != F4L_ICP_POINT2PLANE) return F4L_EINVAL;
    if (search_precision != F4L_SEARCH_F32 && search_precision != F4L_SEARCH_F64) return F4L_EINVAL;
    if (mode == F4L_ICP_POINT2PLANE && max_tgt_patch_host > 0 && !tgt_normals) return F4L_EINVAL;
    if ((max_src_patch_host > 0 && !src) || (max_tgt_patch_host > 0 && !tgt)) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_src_patch_host > 0x3fffffffLL || max_tgt_patch_host > 0x3fffffffLL)
        return F4L_EUNSUPPORTED;
    const bool f64 = search_precision == F4L_SEARCH_F64;
    const int cap_max = f64 ? ICP_LDS_TGT_MAX * 2 / 3 : ICP_LDS_TGT_MAX;  // 24 B vs 16 B per staged target
    IcpArgs a;
    a.src = src; a.src_off = src_off; a.tgt = tgt; a.tgt_off = tgt_off; a.P = P;
    a.init_T = init_T; a.tgt_normals = tgt_normals;
    a.r2 = max_corr_dist > 0.0 ? max_corr_dist * max_corr_dist : 0.0;
    a.max_iter = max_iter; a.rel_fitness = rel_fitness; a.rel_rmse = rel_rmse; a.fixed_iters = fixed_iters;
    a.lds_cap = (int)(max_tgt_patch_host < cap_max ? max_tgt_patch_host : cap_max);
    { const char *dbg = getenv("F4L_ICP_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
    a.T_out = T_out; a.fitness_out = fitness_out; a.rmse_out = rmse_out; a.iters_out = iters_out; a.corr_out = corr_out;
    const size_t lds = (size_t)(ICP_NW * 32 + 16) * sizeof(double) + (size_t)a.lds_cap * (f64 ? 24 : 16);
    const bool two = max_src_patch_host > ICP_NT;  // two source points per lane once patches exceed one pass
    return f64 ? launch_icp<double>(a, mode, two, lds, (hipStream_t)stream)
               : launch_icp<float>(a, mode, two, lds, (hipStream_t)stream);
}
