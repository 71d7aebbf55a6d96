// patch_ops.hip -- per-patch helpers around the ICP loop.
//
//  * patch_normals_kernel : `pcd.estimate_normals()` of utils/o3d_tools.py:29-30 for every patch cloud at once
//    (Open3D default KDTreeSearchParamKNN(knn=30), fast_normal_computation=True): exact kNN inside the
//    patch with the wave-resident top-k of topk.h, covariance from cumulants in double, unit eigenvector of
//    the smallest eigenvalue via the non-iterative symmetric 3x3 solver, (0,0,1) when degenerate.
//  * nn_refine_kernel : `refine_dvfs_with_threshold` of src/coarse_to_fine_matching_base.py:48-97 (a per-point
//    Python loop over an Open3D KD-tree in the reference) for every patch at once.
//
// Both stage the patch once in LDS (coalesced dword loads of the packed [n][3] floats) and then only
// broadcast-read it.
#include "f4l_device.h"
#include "patch_grid.h"
#include "topk.h"
#include "lane_topk.h"

namespace f4l {

constexpr int PN_NW = 4;
constexpr int PN_NT = PN_NW * 64;
constexpr int PN_LDS_MAX = 12288;  // points of a patch kept in LDS (12 B each = 144 KiB)

__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// eigenvector of the symmetric A = (a00,a01,a02,a11,a12,a22) for a simple eigenvalue ev: the best conditioned
// cross product of two rows of A - ev I
__device__ __forceinline__ void eigvec0(const double *A, double ev, double *out) {
    const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01);
    cross3(r0, r2, c02);
    cross3(r1, r2, c12);
    const double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
    double bx = c01[0], by = c01[1], bz = c01[2], dm = d0;
    if (d1 > dm) { dm = d1; bx = c02[0]; by = c02[1]; bz = c02[2]; }
    if (d2 > dm) { dm = d2; bx = c12[0]; by = c12[1]; bz = c12[2]; }
    if (dm > 0.0) {
        const double s = 1.0 / sqrt(dm);
        out[0] = bx * s; out[1] = by * s; out[2] = bz * s;
    } else {
        out[0] = out[1] = out[2] = 0.0;
    }
}

// eigenvector for ev1 inside the plane orthogonal to e0
__device__ __forceinline__ void eigvec1(const double *A, const double *e0, double ev1, double *out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        const double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0.0; U[2] = e0[0] * inv;
    } else {
        const double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0.0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3(e0, U, V);
    const double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                          A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    const double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                          A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = dot3(U, AU) - ev1, m01 = dot3(U, AV), m11 = dot3(V, AV) - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0.0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i] = U[i];
        }
    } else {
        if (fmax(a11, a01) > 0.0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i] = U[i];
        }
    }
}

// unit eigenvector of the smallest eigenvalue of symmetric C (c00,c01,c02,c11,c12,c22); zero vector if C == 0
// the normal of point i: as double where the caller asked for doubles (Open3D keeps its normals in double and registration_icp
// reads them so), else rounded to float32
__device__ __forceinline__ void store_normal(float *__restrict__ f32, double *__restrict__ f64, int64_t i, const double *nv) {
    if (f64) { f64[3 * i] = nv[0]; f64[3 * i + 1] = nv[1]; f64[3 * i + 2] = nv[2]; }
    else { f32[3 * i] = (float)nv[0]; f32[3 * i + 1] = (float)nv[1]; f32[3 * i + 2] = (float)nv[2]; }
}
__device__ __forceinline__ void smallest_eigvec3(const double *C, double *out) {
    double mx = C[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) mx = C[i] > mx ? C[i] : mx;
    if (mx == 0.0) { out[0] = out[1] = out[2] = 0.0; return; }
    double A[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[i] = C[i] / mx;
    const double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0.0) {
        const double q = (A[0] + A[3] + A[5]) / 3.0;
        const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        const double c00 = b11 * b22 - A[4] * A[4];
        const double c01 = A[1] * b22 - A[4] * A[2];
        const double c02 = A[1] * A[4] - b11 * A[2];
        const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double half_det = det * 0.5;
        half_det = half_det < -1.0 ? -1.0 : (half_det > 1.0 ? 1.0 : half_det);
        const double angle = acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = cos(angle) * 2.0;
        const double beta0 = cos(angle + two_thirds_pi) * 2.0;
        const double beta1 = -(beta0 + beta2);
        const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
        double e0[3], e1[3], e2[3];
        if (half_det >= 0.0) {
            eigvec0(A, ev2, e2);
            if (ev2 < ev0 && ev2 < ev1) { out[0] = e2[0]; out[1] = e2[1]; out[2] = e2[2]; return; }
            eigvec1(A, e2, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { out[0] = e1[0]; out[1] = e1[1]; out[2] = e1[2]; return; }
            cross3(e1, e2, out);
        } else {
            eigvec0(A, ev0, e0);
            if (ev0 < ev1 && ev0 < ev2) { out[0] = e0[0]; out[1] = e0[1]; out[2] = e0[2]; return; }
            eigvec1(A, e0, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) { out[0] = e1[0]; out[1] = e1[1]; out[2] = e1[2]; return; }
            cross3(e0, e1, out);
        }
    } else {
        out[0] = out[1] = out[2] = 0.0;
        if (A[0] < A[3] && A[0] < A[5]) out[0] = 1.0;
        else if (A[3] < A[0] && A[3] < A[5]) out[1] = 1.0;
        else out[2] = 1.0;
    }
}

// The covariance of a point's k neighbours (itself included) and its smallest eigenvector: Open3D's ComputeCovariance sums the
// cumulants in the order of the search result (ascending distance), and so does every kernel here -- one lane walking its sorted
// list, or a wave reading its sorted slots lane by lane -- so a query's normal does not depend on which kernel or path took it
// (round 6: the wave paths used a butterfly sum before, equal to the last bits only).
__device__ __forceinline__ void cov_add(double (&cum)[9], double a, double b, double c) {
    cum[0] += a; cum[1] += b; cum[2] += c;
    cum[3] += a * a; cum[4] += a * b; cum[5] += a * c; cum[6] += b * b; cum[7] += b * c; cum[8] += c * c;
}
__device__ __forceinline__ void cov_normal(double (&cum)[9], int k, double (&nv)[3]) {
    if (k < 3) { nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0; return; }  // identity covariance -> no preferred axis -> (0,0,1)
    const double ik = 1.0 / (double)k;
#pragma unroll
    for (int i = 0; i < 9; ++i) cum[i] *= ik;
    double Cm[6];
    Cm[0] = cum[3] - cum[0] * cum[0];
    Cm[1] = cum[4] - cum[0] * cum[1];
    Cm[2] = cum[5] - cum[0] * cum[2];
    Cm[3] = cum[6] - cum[1] * cum[1];
    Cm[4] = cum[7] - cum[1] * cum[2];
    Cm[5] = cum[8] - cum[2] * cum[2];
    smallest_eigvec3(Cm, nv);
    if (nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] == 0.0) { nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0; }
}
// (a wave's sorted slots: lane j holds the j-th neighbour's coordinates)
__device__ __forceinline__ void wave_covariance_normal(double a, double b, double c, int k, double (&nv)[3]) {
    double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) cov_add(cum, readlane_f64(a, j), readlane_f64(b, j), readlane_f64(c, j));
    cov_normal(cum, k, nv);
}

// One workgroup per patch, one wave per query point (queries strided over the 4 waves).
__global__ __launch_bounds__(PN_NT) void patch_normals_kernel(const float *__restrict__ pts,
                                                               const int64_t *__restrict__ off, int64_t P, int knn,
                                                               int lds_cap, int min_n, float *__restrict__ normals,
                                                               double *__restrict__ normals64) {
    extern __shared__ __attribute__((aligned(16))) float pl[];  // packed xyz of the patch
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = off[p];
    const int n = (int)(off[p + 1] - o);
    if (n == 0 || n <= min_n) return;  // (patches up to min_n points: patch_normals_lanes_kernel)
    const float *__restrict__ pg = pts + 3 * o;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool in_lds = n <= lds_cap;
    if (in_lds) {
        for (int i = tid; i < 3 * n; i += PN_NT) pl[i] = pg[i];
        __syncthreads();
    }
    const float *__restrict__ base = in_lds ? pl : pg;
    const int k = knn < n ? knn : n;
    for (int q = wave; q < n; q += PN_NW) {
        const float qx = base[3 * q], qy = base[3 * q + 1], qz = base[3 * q + 2];
        WaveTopK best;
        best.reset();
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int c = c0 + lane;
            double cd = __builtin_inf();
            int ci = 0x7fffffff;
            if (c < n) {
                cd = dist2_exact(base[3 * c], base[3 * c + 1], base[3 * c + 2], qx, qy, qz);
                ci = c;
            }
            if (c0 == 0) best.fill_sorted(cd, ci);  // first batch: sort in place instead of 64 insertions
            else best.offer(cd, ci, k);
        }
        // the k neighbours (self included) in the order of the list
        double nv[3];
        {
            const bool have = lane < k;
            const int j = have ? best.i : q;
            wave_covariance_normal((double)base[3 * j], (double)base[3 * j + 1], (double)base[3 * j + 2], k, nv);
        }
        if (lane == 0) {
            store_normal(normals, normals64, o + q, nv);
        }
    }
}


// ---- the same with one LANE per query (patches of up to PL_MAX_N points, k <= PL_MAX_K) ------------------------------------
// The kernel above spends ~100 serial top-k insertions of a whole wave on every query (12 ms per 1 M points in patches of
// 500).  Here a lane owns a query and all lanes of a wave walk the patch's points together (broadcast reads of the SoA copy
// in LDS), the way knn_lanes_kernel walks a block's candidates: pass 1 bins the approximate d2 into a per-lane histogram
// (32 quarter-octave bins below a top derived from the patch's point density), the bin at which the count reaches k gives
// the threshold, pass 2 collects the candidates below it (widened far beyond the float32 error), the <= 43 survivors are
// measured exactly and sorted on registers, the covariance of the first k is summed in neighbour order (the order of
// Open3D's ComputeCovariance) and every lane solves its own 3 x 3 eigenproblem.  A query whose survivors overflow the
// list is redone by the wave with the exact top-k of the kernel above.
constexpr int PL_NW = 4, PL_NT = PL_NW * 64, PL_NB = 32, PL_CAP = 43, PL_MAX_K = 36, PL_MAX_N = 8192;
constexpr float PL_FAR = 1e30f;  // (1e30^2 overflows to +inf)
typedef float pl_f2 __attribute__((ext_vector_type(2)));  // (two candidates per v_pk_* instruction)
// Workgroups per CU the compiler is asked to fit: 3 = 168 VGPRs, three waves per SIMD, ~40 dwords of the register sort spilled --
// 30.4 ms at C4 against 34.3 ms with the 223 VGPRs / two waves per SIMD it takes unasked (round 5).
#ifndef PL_MIN_WGS
#define PL_MIN_WGS 3
#endif
// (time-only ablations for tools/build_variant.sh: results are wrong with any of them set)
#ifdef PL_ABL_NOPASS1
#define PL_ABL_N1(n) ((n) < 64 ? (n) : 64)
#else
#define PL_ABL_N1(n) (n)
#endif
#ifdef PL_ABL_NOPASS2
#define PL_ABL_N2(n) ((n) < 40 ? (n) : 40)
#else
#define PL_ABL_N2(n) (n)
#endif
__device__ __forceinline__ void pl_covariance_normal(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                     const int (&pay)[PL_CAP], int k, double (&nv)[3]) {
    double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < PL_MAX_K; ++j)
        if (j < k) {
            const int id = pay[j];
            cov_add(cum, (double)x[id], (double)y[id], (double)z[id]);
        }
    cov_normal(cum, k, nv);
}
__global__ __launch_bounds__(PL_NT, PL_MIN_WGS) void patch_normals_lanes_kernel(const float *__restrict__ pts, const int64_t *__restrict__ off,
                                                                       int64_t P, int knn, int cap_pad, float *__restrict__ normals,
                                                                       double *__restrict__ normals64) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    float *xs = reinterpret_cast<float *>(pl_smem), *ys = xs + cap_pad, *zs = ys + cap_pad;  // the patch, SoA (original coordinates)
    unsigned int *hist_all = reinterpret_cast<unsigned int *>(zs + cap_pad);                 // [PL_NW][PL_NB / 2][64]: two 16-bit bins per word
    unsigned short *list_all = reinterpret_cast<unsigned short *>(hist_all + PL_NW * (PL_NB / 2) * 64);  // [PL_NW][PL_CAP + 1][64]
    __shared__ float s_box[PL_NW][6];
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = off[p];
    const int n = (int)(off[p + 1] - o);
    if (n == 0 || n > cap_pad - 8) return;  // (larger patches: patch_normals_kernel)
    const float *__restrict__ pg = pts + 3 * o;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned int *hist = hist_all + wave * (PL_NB / 2) * 64 + lane;
    unsigned short *list = list_all + wave * (PL_CAP + 1) * 64 + lane;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const int n_pad = (n + 7) & ~7;
    for (int i = tid; i < n_pad; i += PL_NT) {
        const bool real = i < n;
        // (places past the patch hold a point at infinity: d2 = +inf falls into the last bin of pass 1 -- which a threshold only
        //  reaches when it takes every point anyway -- and below no edge in pass 2, so neither pass asks which candidates are real)
        const float a = real ? pg[3 * i] : PL_FAR, b = real ? pg[3 * i + 1] : PL_FAR, c = real ? pg[3 * i + 2] : PL_FAR;
        xs[i] = a; ys[i] = b; zs[i] = c;
        if (real) {
            mn[0] = fminf(mn[0], a); mn[1] = fminf(mn[1], b); mn[2] = fminf(mn[2], c);
            mx[0] = fmaxf(mx[0], a); mx[1] = fmaxf(mx[1], b); mx[2] = fmaxf(mx[2], c);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], m, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], m, 64));
        }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { s_box[wave][d] = mn[d]; s_box[wave][3 + d] = mx[d]; }
    }
    __syncthreads();
    const int k = knn < n ? knn : n;
    // top of the histogram's range: 16 x the squared radius that holds k points at the patch's mean density (surface: the two
    // largest extents of the bounding box; a degenerate box: its diagonal)
    int bin_base;
    float slack;
    {
        float e[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float lo = s_box[0][d], hi = s_box[0][3 + d];
#pragma unroll
            for (int w = 1; w < PL_NW; ++w) { lo = fminf(lo, s_box[w][d]); hi = fmaxf(hi, s_box[w][3 + d]); }
            e[d] = hi - lo;
        }
        const float diag2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        const float e_min = fminf(e[0], fminf(e[1], e[2]));
        const float area = e[0] * e[1] * e[2] > 0.f ? (e[0] * e[1] * e[2]) / e_min : 0.f;  // product of the two largest
        float top = area > 0.f ? 16.f * (float)k * area / (3.14159265f * (float)n) : diag2;
        top = top < diag2 ? top : diag2;
        top = top > 1e-30f ? top : 1e-30f;
        bin_base = ((int)(__float_as_uint(top) >> 21) - (PL_NB - 1)) & ~1;  // (top >= 1e-30: positive; even: a counter word = two bins)
        slack = 1e-6f * top;
    }
    const float range_lo = __uint_as_float((unsigned int)bin_base << 21), range_hi = __uint_as_float((((unsigned int)bin_base + PL_NB) << 21) - 1u);
    // (the lane's counters as an LDS address, shifted so that + (bits of d2 >> 22) << 8 is the counter's word)
    const unsigned int hist_addr = (unsigned int)(size_t)((__attribute__((address_space(3))) unsigned int *)hist) - (((unsigned int)bin_base >> 1) << 8);
    for (int q0 = 0; q0 < n; q0 += PL_NT) {  // (whole waves iterate together)
        const int q = q0 + tid;
        const bool valid = q < n;
        const float qx = xs[valid ? q : 0], qy = ys[valid ? q : 0], qz = zs[valid ? q : 0];
#pragma unroll
        for (int b = 0; b < PL_NB / 2; ++b) hist[b * 64] = 0u;
        // pass 1: per-lane histogram of the approximate d2 (differences of nearby float coordinates are exact or nearly so), two
        // candidates per packed instruction.  d2 is clamped as a FLOAT to the histogram's range, so its bits give the counter's
        // word (bits 22.. : bin_base is even) and half (bit 21) directly: 5 integer instructions per candidate where clamping the bin
        // index took 8 (round 6).  (Lanes past the patch's last query count too: nobody reads their counters.)
        // (reading the next eight candidates ahead of this step's atomics, the way knn_lanes_kernel does, was measured: 36.9 ms
        //  against 30.3 at C4 -- three waves per SIMD hide the LDS latency already and the second register set costs more)
        const pl_f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        for (int c = 0; c < PL_ABL_N1(n_pad); c += 8) {
            const float4 xa = *reinterpret_cast<const float4 *>(xs + c), xb = *reinterpret_cast<const float4 *>(xs + c + 4);
            const float4 ya = *reinterpret_cast<const float4 *>(ys + c), yb = *reinterpret_cast<const float4 *>(ys + c + 4);
            const float4 za = *reinterpret_cast<const float4 *>(zs + c), zb = *reinterpret_cast<const float4 *>(zs + c + 4);
            const pl_f2 cx[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}}, cy[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}},
                        cz[4] = {{za.x, za.y}, {za.z, za.w}, {zb.x, zb.y}, {zb.z, zb.w}};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const pl_f2 dx = cx[w] - qx2, dy = cy[w] - qy2, dz = cz[w] - qz2;
                const pl_f2 d2 = dx * dx + dy * dy + dz * dz;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float t = __builtin_amdgcn_fmed3f(d2[h], range_lo, range_hi);
                    unsigned int at, by;
                    asm("v_lshrrev_b32 %0, 22, %1\n\tv_lshl_add_u32 %0, %0, 8, %2" : "=&v"(at) : "v"(t), "v"(hist_addr));
                    asm("v_bfe_u32 %0, %1, 21, 1\n\tv_mad_u32_u24 %0, %0, %2, 1" : "=&v"(by) : "v"(t), "s"(65535u));
                    __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)(size_t)at, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        // the first bin at which the count reaches k
        int T = PL_NB;
        unsigned int cum = 0u;
#pragma unroll
        for (int b = 0; b < PL_NB; ++b) {
            const unsigned int h = (hist[(b >> 1) * 64] >> ((b & 1) * 16)) & 0xffffu;
            if (T == PL_NB && cum + h >= (unsigned int)k) T = b;
            cum += h;
        }
        bool fb = valid && T == PL_NB;  // (cannot happen: every point of the patch lands in a bin)
        float edge = T >= PL_NB - 1 ? __builtin_inff() : __uint_as_float((unsigned int)(T + bin_base + 1) << 21) * 1.0001f + slack;
        edge = valid ? edge : -1.0f;
        // pass 2: the candidates below the threshold, 32 at a time as a bit mask (a compare and an add-with-carry per candidate), and
        // the few set bits of a mask go to the lane's list (round 6; before: a conditional list write per candidate, 15 instructions)
        int cnt = 0;
        for (int c = 0; c < PL_ABL_N2(n_pad); c += 32) {
            unsigned int mask = 0u;
#pragma unroll
            for (int g = 0; g < 32; g += 8) {
                if (c + g < n_pad) {  // (uniform)
                    const float4 xa = *reinterpret_cast<const float4 *>(xs + c + g), xb = *reinterpret_cast<const float4 *>(xs + c + g + 4);
                    const float4 ya = *reinterpret_cast<const float4 *>(ys + c + g), yb = *reinterpret_cast<const float4 *>(ys + c + g + 4);
                    const float4 za = *reinterpret_cast<const float4 *>(zs + c + g), zb = *reinterpret_cast<const float4 *>(zs + c + g + 4);
                    const pl_f2 cx[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}}, cy[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}},
                                cz[4] = {{za.x, za.y}, {za.z, za.w}, {zb.x, zb.y}, {zb.z, zb.w}};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const pl_f2 dx = cx[w] - qx2, dy = cy[w] - qy2, dz = cz[w] - qz2;
                        const pl_f2 d2 = dx * dx + dy * dy + dz * dz;
                        // mask = 2 * mask + (d2 < edge): candidate c + j of the block ends at bit 31 - j
                        asm("v_cmp_lt_f32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                            "v_cmp_lt_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                            : "+v"(mask) : "v"(d2[0]), "v"(d2[1]), "v"(edge) : "vcc");
                    }
                } else {
                    mask <<= 8;
                }
            }
            while (mask != 0u) {
                const int j = __clz((int)mask);
                mask &= ~(0x80000000u >> j);
                list[(cnt < PL_CAP ? cnt : PL_CAP) * 64] = (unsigned short)(c + j);
                ++cnt;
            }
        }
        if (cnt > PL_CAP) { fb = true; cnt = PL_CAP; }
#if defined(PL_ABL_NOPASS1) || defined(PL_ABL_NOPASS2)
        fb = false;
#endif
        // the survivors, exact d2 from the float coordinates, sorted on registers
        double key[PL_CAP];
        int pay[PL_CAP];
#pragma unroll
        for (int j = 0; j < PL_CAP; ++j) {
            const bool ok = j < cnt;
            const int id = ok ? (int)list[j * 64] : 0;
            key[j] = ok ? dist2_exact(xs[id], ys[id], zs[id], qx, qy, qz) : __builtin_inf();
            pay[j] = ok ? id : 0x7fffffff;
        }
#ifndef PL_ABL_NOSORT
        lane_sort_ascending<PL_CAP>(key, pay);
        lane_order_ties<PL_CAP>(key, pay);
#endif
        double nv[3];
#ifdef PL_ABL_NOEIG
        nv[0] = key[0] + key[PL_CAP - 1] + key[20]; nv[1] = pay[0] + pay[PL_CAP - 1] + pay[20]; nv[2] = 1.0;
#else
        pl_covariance_normal(xs, ys, zs, pay, k, nv);
#endif
        if (valid && !fb) {
            store_normal(normals, normals64, o + q, nv);
        }
        // the rare query whose survivors overflowed the list: the wave's exact top-k (patch_normals_kernel's arithmetic)
        unsigned long long redo = __ballot(fb);
        while (redo != 0ULL) {
            const int src_lane = __ffsll((long long)redo) - 1;
            redo &= redo - 1ULL;
            const int rq = q0 + wave * 64 + src_lane;
            const float rx = xs[rq], ry = ys[rq], rz = zs[rq];
            WaveTopK best;
            best.reset();
            for (int c0 = 0; c0 < n; c0 += 64) {
                const int c = c0 + lane;
                double cd = __builtin_inf();
                int ci = 0x7fffffff;
                if (c < n) { cd = dist2_exact(xs[c], ys[c], zs[c], rx, ry, rz); ci = c; }
                if (c0 == 0) best.fill_sorted(cd, ci);
                else best.offer(cd, ci, k);
            }
            double rn[3];
            {
                const bool have = lane < k;
                const int j = have ? best.i : rq;
                wave_covariance_normal((double)xs[j], (double)ys[j], (double)zs[j], k, rn);
            }
            if (lane == 0) {
                store_normal(normals, normals64, o + rq, rn);
            }
        }
    }
}

// refine_dvfs_with_threshold: one workgroup per patch.  The target patch is counting-sorted into the LDS-resident
// uniform grid of patch_grid.h (cell edge >= the patch's threshold), every lane owns source points and asks the
// grid for the nearest target within the threshold: the minimiser of (d2, index), i.e. what a scan in index
// order returns.  Patches beyond the LDS budget fall back to that scan over global memory.
struct NnRefineArgs {
    const float *src;
    const int64_t *src_off;
    const float *tgt;
    const int64_t *tgt_off;
    int64_t P;
    const double *T;
    const double *thr;
    int tgt_cap, cell_cap;
    int32_t *nn_out;
    float *out6;
    int skip_nt = -1;  // nn_refine_kernel leaves patches of up to this many targets to nn_refine_small_kernel (-1: none)
};

__global__ __launch_bounds__(PN_NT) void nn_refine_kernel(NnRefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nr_smem[];
    float *red = reinterpret_cast<float *>(nr_smem);  // PN_NW * 8 floats of build scratch
    GridPt<float> *tl = reinterpret_cast<GridPt<float> *>(red + PN_NW * 8);
    unsigned int *rl = reinterpret_cast<unsigned int *>(tl + a.tgt_cap + 1);
    unsigned short *E = reinterpret_cast<unsigned short *>(rl + (GRID_ROWS + 1) * PN_NT);
    const int64_t p = blockIdx.x;
    if (p >= a.P) return;
    const int64_t s0 = a.src_off[p], t0 = a.tgt_off[p];
    const int ns = (int)(a.src_off[p + 1] - s0), nt = (int)(a.tgt_off[p + 1] - t0);
    if (nt <= a.skip_nt) return;  // (uniform: before any barrier)
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    const int tid = (int)threadIdx.x;
    const double th = a.thr[p];
    const float th2 = (float)(th * th);
    const bool searchable = nt > 0 && th > 0.0;
    const bool in_lds = searchable && nt <= a.tgt_cap;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    PatchGrid<float> g;
    g.minx = g.miny = g.minz = 0.f; g.h = 1.f; g.inv_h = 1.f; g.inv_hx = 1.f; g.inv_hz = 1.f; g.nx = g.ny = g.nz = 1; g.xs = 1; g.wmax = 1;
    if (in_lds) grid_build<float, PN_NT>(tg, nt, ox, oy, oz, (float)th * 1.000001f, a.cell_cap, tl, E, red, g);
    const double *Tp = a.T + 16 * p;
    // p' = R (s' + o) + t - o with s' = s - o
    float Rf[9], tf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Rf[3 * i] = (float)Tp[4 * i]; Rf[3 * i + 1] = (float)Tp[4 * i + 1]; Rf[3 * i + 2] = (float)Tp[4 * i + 2];
        const double o_i = i == 0 ? ox : (i == 1 ? oy : oz);
        tf[i] = (float)(Tp[4 * i] * (double)ox + Tp[4 * i + 1] * (double)oy + Tp[4 * i + 2] * (double)oz + Tp[4 * i + 3] - o_i);
    }
    for (int base = 0; base < ns; base += PN_NT) {  // whole waves take part in every query (grid_nn is wave wide)
        const int i = base + tid;
        const bool valid = i < ns;
        const int ii = valid ? i : ns - 1;
        const float sx = sg[3 * ii], sy = sg[3 * ii + 1], sz = sg[3 * ii + 2];
        const float x = sx - ox, y = sy - oy, z = sz - oz;
        const float px = Rf[0] * x + Rf[1] * y + Rf[2] * z + tf[0];
        const float py = Rf[3] * x + Rf[4] * y + Rf[5] * z + tf[1];
        const float pz = Rf[6] * x + Rf[7] * y + Rf[8] * z + tf[2];
        Best<float> best;
        best.init(th2);
        int bj = -1;
        if (in_lds) {
            grid_nn<float, PN_NT>(g, tl, nt, E, rl, valid, px, py, pz, best);
            if (best.found()) bj = best.id();
        } else if (searchable) {
            for (int j = 0; j < nt; ++j)
                best.offer(grid_d2(px - (tg[3 * j] - ox), py - (tg[3 * j + 1] - oy), pz - (tg[3 * j + 2] - oz)), (unsigned int)j);
            if (best.found()) bj = (int)best.tag();
        }
        const bool hit = valid && bj >= 0 && best.d2() < th2;  // :80 dists[0] < distance_threshold ** 2
        if (valid) {
            if (a.nn_out) a.nn_out[s0 + i] = hit ? bj : -1;
            if (a.out6) {
                float *o6 = a.out6 + 6 * (s0 + i);
                o6[0] = sx; o6[1] = sy; o6[2] = sz;
                o6[3] = hit ? tg[3 * bj] : 0.f; o6[4] = hit ? tg[3 * bj + 1] : 0.f; o6[5] = hit ? tg[3 * bj + 2] : 0.f;
            }
        }
    }
}

// Small target patches (supervoxels: a few dozen points -- the patches of a tile's full path): a 256-thread workgroup per patch
// keeps one wave in four busy and spends more on building a grid than the grid saves.  Here a WAVE takes a patch, stages the target
// patch in LDS the way the scan above reads it (tg - origin, float32) and every lane measures ALL targets for its source point:
// the minimiser of (d2, index) below the threshold, the same values and the same winner as the grid's search and as the scan
// (f4l_nn_refine at 100 M points in 1.67 M supervoxels: 12.7 -> 3 ms).
constexpr int NRS_CAP = 128;
__global__ __launch_bounds__(256) void nn_refine_small_kernel(NnRefineArgs a) {
    __shared__ float4 s_t[4][NRS_CAP];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int64_t p = (int64_t)blockIdx.x * 4 + wave;
    if (p >= a.P) return;  // (no workgroup barrier below: waves are on their own)
    const int64_t s0 = a.src_off[p], t0 = a.tgt_off[p];
    const int ns = (int)(a.src_off[p + 1] - s0), nt = (int)(a.tgt_off[p + 1] - t0);
    if (nt > NRS_CAP) return;  // (nn_refine_kernel's)
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    const double th = a.thr[p];
    const float th2 = (float)(th * th);
    const bool searchable = nt > 0 && th > 0.0;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    float4 *tw = s_t[wave];
    for (int j = lane; j < nt; j += 64) tw[j] = make_float4(tg[3 * j] - ox, tg[3 * j + 1] - oy, tg[3 * j + 2] - oz, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double *Tp = a.T + 16 * p;
    float Rf[9], tf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Rf[3 * i] = (float)Tp[4 * i]; Rf[3 * i + 1] = (float)Tp[4 * i + 1]; Rf[3 * i + 2] = (float)Tp[4 * i + 2];
        const double o_i = i == 0 ? ox : (i == 1 ? oy : oz);
        tf[i] = (float)(Tp[4 * i] * (double)ox + Tp[4 * i + 1] * (double)oy + Tp[4 * i + 2] * (double)oz + Tp[4 * i + 3] - o_i);
    }
    for (int base = 0; base < ns; base += 64) {
        const int i = base + lane;
        const bool valid = i < ns;
        const int ii = valid ? i : ns - 1;
        const float sx = sg[3 * ii], sy = sg[3 * ii + 1], sz = sg[3 * ii + 2];
        const float x = sx - ox, y = sy - oy, z = sz - oz;
        const float px = Rf[0] * x + Rf[1] * y + Rf[2] * z + tf[0];
        const float py = Rf[3] * x + Rf[4] * y + Rf[5] * z + tf[1];
        const float pz = Rf[6] * x + Rf[7] * y + Rf[8] * z + tf[2];
        Best<float> best;
        best.init(th2);
        if (searchable) {
#pragma unroll 4
            for (int j = 0; j < nt; ++j) {
                const float4 t = tw[j];
                best.offer(grid_d2(px - t.x, py - t.y, pz - t.z), (unsigned int)j);
            }
        }
        const int bj = best.found() ? (int)best.tag() : -1;
        const bool hit = valid && bj >= 0 && best.d2() < th2;  // :80 dists[0] < distance_threshold ** 2
        if (valid) {
            if (a.nn_out) a.nn_out[s0 + i] = hit ? bj : -1;
            if (a.out6) {
                float *o6 = a.out6 + 6 * (s0 + i);
                o6[0] = sx; o6[1] = sy; o6[2] = sz;
                o6[3] = hit ? tg[3 * bj] : 0.f; o6[4] = hit ? tg[3 * bj + 1] : 0.f; o6[5] = hit ? tg[3 * bj + 2] : 0.f;
            }
        }
    }
}

}  // namespace f4l

static int patch_normals_launch(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                                float *normals_out, double *normals64_out, void *stream) {
    using namespace f4l;
    if (P < 0 || !off || knn < 1 || max_patch_host < 0 || (max_patch_host > 0 && (!pts || (!normals_out && !normals64_out)))) return F4L_EINVAL;
    if (knn > F4L_MAX_K) return F4L_EUNSUPPORTED;
    if (P == 0 || max_patch_host == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_patch_host > 0x3fffffffLL) return F4L_EUNSUPPORTED;
    // patches of up to PL_MAX_N points (k <= PL_MAX_K): one lane per query; larger ones: one wave per query
    int min_n = 0;
    if (knn <= PL_MAX_K && !getenv("F4L_PATCH_NORMALS_WAVES")) {
        const int big = (int)(max_patch_host < PL_MAX_N ? max_patch_host : PL_MAX_N);
        const int cap_pad = ((big + 7) & ~7) + 8;
        const size_t lds = (size_t)cap_pad * 12 + (size_t)PL_NW * (PL_NB / 2) * 64 * 4 + (size_t)PL_NW * (PL_CAP + 1) * 64 * 2;
        if (lds > 64 * 1024)
            F4L_HIP_CHECK(hipFuncSetAttribute((const void *)patch_normals_lanes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(patch_normals_lanes_kernel, dim3((unsigned)P), dim3(PL_NT), lds, (hipStream_t)stream, pts, off, P, knn, cap_pad, normals_out, normals64_out);
        F4L_LAUNCH_CHECK();
        min_n = big;
        if (max_patch_host <= big) return F4L_OK;
    }
    const int cap = (int)(max_patch_host < PN_LDS_MAX ? max_patch_host : PN_LDS_MAX);
    const size_t lds = (size_t)cap * 12;
    if (lds > 64 * 1024)
        F4L_HIP_CHECK(hipFuncSetAttribute((const void *)patch_normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(patch_normals_kernel, dim3((unsigned)P), dim3(PN_NT), lds, (hipStream_t)stream, pts, off, P, knn,
                       cap, min_n, normals_out, normals64_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

extern "C" int f4l_patch_normals(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                                 float *normals_out, void *stream) {
    return patch_normals_launch(pts, off, P, knn, max_patch_host, normals_out, nullptr, stream);
}
// The same normals as doubles: what Open3D's `estimate_normals()` leaves in the cloud and `registration_icp` reads
// (utils/o3d_tools.py:29-30, 46-50); pass them to f4l_piecewise_icp / f4l_patch_loop with F4L_ICP_NORMALS_F64.
extern "C" int f4l_patch_normals_f64(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                                     double *normals_out, void *stream) {
    return patch_normals_launch(pts, off, P, knn, max_patch_host, nullptr, normals_out, stream);
}

extern "C" int f4l_nn_refine(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                             const double *T, const double *thr, int64_t max_tgt_patch_host, int32_t *nn_out, float *out6,
                             void *stream) {
    using namespace f4l;
    if (P < 0 || !src_off || !tgt_off || !T || !thr || max_tgt_patch_host < 0) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_tgt_patch_host > 0x3fffffffLL) return F4L_EUNSUPPORTED;
    int cap = (int)(max_tgt_patch_host < 8192 ? max_tgt_patch_host : 8192);
    if (cap < 1) cap = 1;
    int cells = (int)((2 * (int64_t)cap + 255) & ~(int64_t)255);
    cells = cells < 512 ? 512 : (cells > 16384 ? 16384 : cells);
    const size_t lds = (size_t)PN_NW * 8 * 4 + (size_t)(cap + 1) * 16 + (size_t)(GRID_ROWS + 1) * PN_NT * 4 + ((size_t)cells + 8) * 2;
    if (lds > 64 * 1024)
        F4L_HIP_CHECK(hipFuncSetAttribute((const void *)nn_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NnRefineArgs a;
    a.src = src; a.src_off = src_off; a.tgt = tgt; a.tgt_off = tgt_off; a.P = P; a.T = T; a.thr = thr;
    a.tgt_cap = cap; a.cell_cap = cells; a.nn_out = nn_out; a.out6 = out6;
    // patches of up to NRS_CAP targets: a wave each (nn_refine_small_kernel); the others: a workgroup each, over a grid in LDS
    const bool small = !getenv("F4L_NN_REFINE_NO_SMALL");  // (switch: the test that both give the same answers)
    if (small) {
        hipLaunchKernelGGL(nn_refine_small_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
        F4L_LAUNCH_CHECK();
        // (the grid kernel is launched even when the caller states no patch beyond NRS_CAP: a bound may be understated, and a patch the
        //  wave kernel leaves alone must not keep uninitialised rows -- workgroups of small patches return at once)
        a.skip_nt = NRS_CAP;
    }
    hipLaunchKernelGGL(nn_refine_kernel, dim3((unsigned)P), dim3(PN_NT), lds, (hipStream_t)stream, a);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- f4l_nn_refine's answers as the correspondence lists of f4l_patch_loop ------------------------------------------------------
namespace f4l {
__global__ __launch_bounds__(256) void match_lists_kernel(const float *__restrict__ src, const int64_t *__restrict__ src_off,
                                                          const float *__restrict__ tgt, const int64_t *__restrict__ tgt_off, int64_t P,
                                                          const int32_t *__restrict__ nn, const int64_t *__restrict__ kept_before,
                                                          float *__restrict__ cs, float *__restrict__ ct, int64_t *__restrict__ coff) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // a wave per patch
    if (p >= P) return;
    const int64_t s0 = src_off[p], s1 = src_off[p + 1], t0 = tgt_off[p];
    for (int64_t i = s0 + lane_id(); i < s1; i += 64) {
        const int32_t m = nn[i];
        if (m < 0) continue;
        const int64_t at = kept_before[i], t = t0 + m;
#pragma unroll
        for (int d = 0; d < 3; ++d) { cs[3 * at + d] = src[3 * i + d]; ct[3 * at + d] = tgt[3 * t + d]; }
    }
    if (lane_id() == 0) {
        coff[p] = kept_before[s0];
        if (p == P - 1) coff[P] = kept_before[s1];
    }
}
}  // namespace f4l

extern "C" int f4l_match_lists(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                               const int32_t *nn, const int64_t *kept_before, float *corr_src_out, float *corr_ref_out,
                               int64_t *corr_off_out, void *stream) {
    using namespace f4l;
    if (P < 0 || !src_off || !tgt_off || !corr_off_out || (P > 0 && (!nn || !kept_before))) return F4L_EINVAL;
    if (P == 0) { F4L_HIP_CHECK(hipMemsetAsync(corr_off_out, 0, 8, (hipStream_t)stream)); return F4L_OK; }
    if (P > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipLaunchKernelGGL(match_lists_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, src_off, tgt, tgt_off, P, nn,
                       kept_before, corr_src_out, corr_ref_out, corr_off_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- the loop body's steps before the rigid fit (src/coarse_to_fine_matching_base.py:3254-3320) ------------------------
namespace f4l {

// :3259-3274  `mask = torch.isin(corr[src patch][:, 1], tgt patch)`: for every source point of every patch match, does its
// correspondent lie in the matched target patch?  One thread per source row; membership by binary search in the target
// patch's point ids (ascending, as f4l_labels_to_csr emits them).
__global__ void mutual_mask_kernel(const int64_t *__restrict__ src_ids, const int64_t *__restrict__ src_off,
                                   const int64_t *__restrict__ tgt_ids, const int64_t *__restrict__ tgt_off, int64_t P,
                                   const int64_t *__restrict__ corr_tgt, int64_t n_corr, uint8_t *__restrict__ mask,
                                   int64_t *__restrict__ count) {
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const int64_t s0 = src_off[p], ns = src_off[p + 1] - s0, t0 = tgt_off[p], nt = tgt_off[p + 1] - t0;
        const int64_t *__restrict__ tid = tgt_ids + t0;
        int mine = 0;
        for (int64_t i = threadIdx.x; i < ns; i += blockDim.x) {
            const int64_t s = src_ids[s0 + i];
            const int64_t t = s >= 0 && s < n_corr ? corr_tgt[s] : -1;
            bool in = false;
            if (t >= 0) {
                int64_t lo = 0, hi = nt;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (tid[mid] < t) lo = mid + 1; else hi = mid;
                }
                in = lo < nt && tid[lo] == t;
            }
            mask[s0 + i] = in ? 1 : 0;
            mine += in ? 1 : 0;
        }
        mine = wave_sum(mine);
        __shared__ int s_cnt[16];
        const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
        if (lane_id() == 0) s_cnt[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0 && count) {
            int tot = 0;
            for (int w = 0; w < nw; ++w) tot += s_cnt[w];
            count[p] = tot;
        }
        __syncthreads();
    }
}

// :3304-3320  rigidity of a patch match from its n mutual pairs: |d(s_i, s_j) - d(t_i, t_j)| over all pairs,
//   dist_mean    = sum over i < j / (n (n - 1) / 2)
//   ratio_inlier = (#{(i, j), any order incl. i = j: diff <= thr} - n) / (n (n - 1))
// One workgroup per patch match; distances in double (the reference's float32 torch.cdist is matched to ~1e-6).
// Sets of up to RG_CAP pairs are staged in LDS as doubles and the n (n - 1) / 2 unordered pairs are dealt to the threads one
// evenly -- pair (i, i + s mod n) for the shifts s = 1 .. (n - 1) / 2 (and half the rows of s = n / 2 when n is even), in units
// of one point against eight consecutive shifts: every thread gets the same number of units whatever n is (a thread per ROW
// of the triangle leaves half the lanes idle), the partner points come from LDS (neighbouring lanes, neighbouring words: 6.75
// reads of 8 bytes per pair) and the square roots are v_rsq_f64 + two
// Goldschmidt steps (<= 1 ulp; the distances here are far from the denormal and overflow ranges the IEEE sequence guards).
constexpr int RG_CAP = 1024, RG_B = 8;
__device__ __forceinline__ double rg_sqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-g, g, x);
    g = __builtin_fma(e, h, g);
    return x > 0.0 ? g : 0.0;
}
__device__ __forceinline__ float rg_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // (v_sqrt_f32, 1 ulp)
// R = double: the pair arithmetic of the parity mode.  R = float (f4l_rigidity_check_f32): the differences of two float32
// coordinates of one patch are exact in float32 (both are multiples of the larger one's ulp and the result needs no more bits
// than the operands), so only the three products, their sum and the square root round: a distance is off by <= 3e-7 of
// itself, where the reference's own float32 torch.cdist -- the matrix-product form from 25 points on -- is off by ulp(|x|^2).
// Half the LDS, no double-rate arithmetic; the sums leave a unit of RG_B pairs as doubles.
template <typename R>
__global__ __launch_bounds__(256) void rigidity_kernel(const float *__restrict__ cs, const float *__restrict__ ct,
                                                      const int64_t *__restrict__ off, int64_t P, double thr,
                                                      double *__restrict__ dist_mean, double *__restrict__ ratio_inlier) {
    typedef R R2 __attribute__((ext_vector_type(2)));  // (source, target) side by side: one LDS read, packed float32 arithmetic
    __shared__ R2 pq[3][RG_CAP];
    __shared__ double s_sum[4];
    __shared__ long long s_in[4];
    const R thr_r = (R)thr;
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const int64_t o = off[p];
        const int n = (int)(off[p + 1] - o);
        const float *__restrict__ a = cs + 3 * o, *__restrict__ b = ct + 3 * o;
        double sum = 0.0;
        int inl = 0;  // (a thread sees < 2^31 pairs)
        if (n <= RG_CAP) {
            const bool twice = 2 * n <= RG_CAP;  // the ring unrolled: point j again at j + n, no wrap of i + shift
            for (int t = (int)threadIdx.x; t < 3 * n; t += 256) {
                R2 v;
                v.x = (R)a[t];
                v.y = (R)b[t];
                pq[t % 3][t / 3] = v;
                if (twice) pq[t % 3][t / 3 + n] = v;
            }
            __syncthreads();
            // a unit of work = one point i against RG_B consecutive shifts (its coordinates are read once per unit): units are
            // dealt round-robin, blk * n + i
            const int h = (n - 1) / 2, nblk = (h + RG_B - 1) / RG_B;
            int i = (int)threadIdx.x, blk = 0;
            while (i >= n && blk < nblk) { i -= n; ++blk; }  // (n < 256: this thread's first unit lies in a later block)
            while (blk < nblk) {
                const R2 px = pq[0][i], py = pq[1][i], pz = pq[2][i];
                const int s0 = 1 + blk * RG_B, s1 = s0 + RG_B - 1 < h ? s0 + RG_B - 1 : h;
                R usum = 0;
                if (twice && s1 - s0 + 1 == RG_B) {
                    const R2 *__restrict__ qx = &pq[0][i + s0], *__restrict__ qy = &pq[1][i + s0], *__restrict__ qz = &pq[2][i + s0];
#pragma unroll
                    for (int u = 0; u < RG_B; ++u) {
                        const R2 dx = px - qx[u], dy = py - qy[u], dz = pz - qz[u];
                        const R2 d2 = dx * dx + dy * dy + dz * dz;
                        const R diff = fabs(rg_sqrt(d2.x) - rg_sqrt(d2.y));
                        usum += diff;
                        inl += diff <= thr_r ? 1 : 0;
                    }
                } else {
                    for (int sft = s0; sft <= s1; ++sft) {
                        const int j = i + sft < n ? i + sft : i + sft - n;
                        const R2 dx = px - pq[0][j], dy = py - pq[1][j], dz = pz - pq[2][j];
                        const R2 d2 = dx * dx + dy * dy + dz * dz;
                        const R diff = fabs(rg_sqrt(d2.x) - rg_sqrt(d2.y));
                        usum += diff;
                        inl += diff <= thr_r ? 1 : 0;
                    }
                }
                sum += (double)usum;
                i += 256;
                while (i >= n) { i -= n; ++blk; }
            }
            if ((n & 1) == 0)  // the opposite points of an even ring: each pair once
                for (int r = (int)threadIdx.x; r < n / 2; r += 256) {
                    const int j = r + n / 2;
                    const R2 dx = pq[0][r] - pq[0][j], dy = pq[1][r] - pq[1][j], dz = pq[2][r] - pq[2][j];
                    const R2 d2 = dx * dx + dy * dy + dz * dz;
                    const R diff = fabs(rg_sqrt(d2.x) - rg_sqrt(d2.y));
                    sum += (double)diff;
                    inl += diff <= thr_r ? 1 : 0;
                }
        } else {
            // (sets beyond the LDS capacity: double in both modes)
            // pairs (i, j), i < j, dealt round-robin by row: thread t takes rows t, t + 256, ... (row i has n - 1 - i pairs)
            for (int i = (int)threadIdx.x; i < n; i += 256) {
                const double ax = a[3 * i], ay = a[3 * i + 1], az = a[3 * i + 2], bx = b[3 * i], by = b[3 * i + 1], bz = b[3 * i + 2];
                for (int j = i + 1; j < n; ++j) {
                    const double dx = ax - (double)a[3 * j], dy = ay - (double)a[3 * j + 1], dz = az - (double)a[3 * j + 2];
                    const double ex = bx - (double)b[3 * j], ey = by - (double)b[3 * j + 1], ez = bz - (double)b[3 * j + 2];
                    const double diff = fabs(sqrt(dx * dx + dy * dy + dz * dz) - sqrt(ex * ex + ey * ey + ez * ez));
                    sum += diff;
                    inl += diff <= thr ? 1 : 0;
                }
            }
        }
        sum = wave_sum(sum);
        long long tot_in = (long long)wave_sum(inl);
        if (lane_id() == 0) { s_sum[threadIdx.x >> 6] = sum; s_in[threadIdx.x >> 6] = tot_in; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double pairs = (double)n * (double)(n - 1) / 2.0;
            const double tsum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
            const long long tin = s_in[0] + s_in[1] + s_in[2] + s_in[3];
            // both triangles count in the reference's `sum(diff <= thr)`, the diagonal (always 0 <= thr) is taken off again
            dist_mean[p] = n > 1 ? tsum / pairs : 0.0;
            ratio_inlier[p] = n > 1 ? (double)(2 * tin) / (pairs * 2.0) : 0.0;
        }
        __syncthreads();
    }
}
}  // namespace f4l

extern "C" int f4l_mutual_correspondences(const int64_t *src_ids, const int64_t *src_off, const int64_t *tgt_ids,
                                          const int64_t *tgt_off, int64_t P, const int64_t *corr_tgt, int64_t n_corr,
                                          uint8_t *mask_out, int64_t *count_out, void *stream) {
    if (P < 0 || !src_off || !tgt_off || !mask_out || (P > 0 && (!src_ids || !tgt_ids || !corr_tgt)) || n_corr < 0) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    const unsigned grid = (unsigned)(P < 65535 * 8 ? P : 65535 * 8);
    hipLaunchKernelGGL(f4l::mutual_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src_ids, src_off, tgt_ids, tgt_off, P,
                       corr_tgt, n_corr, mask_out, count_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

namespace f4l {
template <typename R>
static int rigidity_launch(const float *corr_src, const float *corr_ref, const int64_t *corr_off, int64_t P, double thres_dist_diff,
                           double *dist_mean_out, double *ratio_inlier_out, void *stream) {
    if (P < 0 || !corr_off || !dist_mean_out || !ratio_inlier_out || (P > 0 && (!corr_src || !corr_ref))) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    const unsigned grid = (unsigned)(P < 65535 * 8 ? P : 65535 * 8);
    hipLaunchKernelGGL(rigidity_kernel<R>, dim3(grid), dim3(256), 0, (hipStream_t)stream, corr_src, corr_ref, corr_off, P,
                       thres_dist_diff, dist_mean_out, ratio_inlier_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
}  // namespace f4l

extern "C" int f4l_rigidity_check(const float *corr_src, const float *corr_ref, const int64_t *corr_off, int64_t P,
                                  double thres_dist_diff, double *dist_mean_out, double *ratio_inlier_out, void *stream) {
    return f4l::rigidity_launch<double>(corr_src, corr_ref, corr_off, P, thres_dist_diff, dist_mean_out, ratio_inlier_out, stream);
}

extern "C" int f4l_rigidity_check_f32(const float *corr_src, const float *corr_ref, const int64_t *corr_off, int64_t P,
                                      double thres_dist_diff, double *dist_mean_out, double *ratio_inlier_out, void *stream) {
    return f4l::rigidity_launch<float>(corr_src, corr_ref, corr_off, P, thres_dist_diff, dist_mean_out, ratio_inlier_out, stream);
}
