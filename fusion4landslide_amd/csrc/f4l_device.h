// f4l_device.h -- shared device/host helpers for the gfx950 kernels of libf4l_hip.so.
// CDNA4 only: 64-wide wavefronts are assumed everywhere (no warpSize-generic code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f4l.h"

#define F4L_WAVE 64

extern thread_local int f4l_tls_hip_error;

#define F4L_HIP_CHECK(expr)                       \
    do {                                          \
        hipError_t e__ = (expr);                  \
        if (e__ != hipSuccess) {                  \
            f4l_tls_hip_error = (int)e__;         \
            return F4L_EHIP;                      \
        }                                         \
    } while (0)

#define F4L_LAUNCH_CHECK()                        \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) {                  \
            f4l_tls_hip_error = (int)e__;         \
            return F4L_EHIP;                      \
        }                                         \
    } while (0)

namespace f4l {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- wave-level reductions -------------------------------------------------------------------
// xor-butterfly: every lane ends with the full sum.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- DPP reductions (VALU only: no LDS round trips, unlike __shfl which lowers to ds_bpermute) --------
// The classic GCN wave64 pattern: two quad permutes, row_half_mirror, row_mirror, row_bcast15, row_bcast31;
// the total lands in lane 63 and is handed to every lane through an SGPR (v_readlane).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    // bound_ctrl = true: lanes without a source (masked rows keep `old` = 0) read zero
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float bcast_lane63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double bcast_lane63(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the 64 lanes, result uniform in every lane.
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row total
    v += dpp_mov<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave total
    return bcast_lane63(v);
}

// The same for N independent values, stage by stage in groups of at most 6, so that the dependency chains of a
// group interleave without keeping 2 N temporaries alive.
template <typename T, int N, int LO, int HI>
__device__ __forceinline__ void wave_sum_dpp_group(T (&v)[N]) {
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0xB1, 0xf>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0x4E, 0xf>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0x141, 0xf>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0x140, 0xf>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0x142, 0xa>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] += dpp_mov<0x143, 0xc>(v[i]);
#pragma unroll
    for (int i = LO; i < HI; ++i) v[i] = bcast_lane63(v[i]);
}
template <int N, int LO = 0, typename T>
__device__ __forceinline__ void wave_sum_dpp_n(T (&v)[N]) {
    constexpr int HI = LO + 6 < N ? LO + 6 : N;
    wave_sum_dpp_group<T, N, LO, HI>(v);
    if constexpr (HI < N) wave_sum_dpp_n<N, HI, T>(v);
}

// Row sums of NV per-lane values (float or double) without reducing every value on its own: values are taken four at a time and the two
// quad stages TRANSPOSE while they add (a lane keeps the value its position in the quad selects and hands the other
// to its partner), so that after them one register per four values is left; two rotations inside the 16-lane row
// finish the row sum.  On return lane 16 r + 4 q + s holds in x[m] the sum over row r of value 4 m + s (any q), and in
// y[j] the row sum of value 4 (NV / 4) + j.  ~7 instructions per value instead of 18 for six plain butterfly stages;
// the caller adds the rows (through LDS, together with the other waves' rows).
template <int NV, typename T>
__device__ __forceinline__ void row_sums_transposed(const T (&v)[NV], T (&x)[NV / 4 > 0 ? NV / 4 : 1],
                                                    T (&y)[NV % 4 > 0 ? NV % 4 : 1]) {
    const int lane = lane_id();
    const bool p0 = (lane & 1) != 0, p1 = (lane & 2) != 0;
#pragma unroll
    for (int m = 0; m < NV / 4; ++m) {
        const T a0 = v[4 * m], a1 = v[4 * m + 1], a2 = v[4 * m + 2], a3 = v[4 * m + 3];
        const T w0 = (p0 ? a1 : a0) + dpp_mov<0xB1, 0xf>(p0 ? a0 : a1);  // pair sum of value 4 m + p0
        const T w1 = (p0 ? a3 : a2) + dpp_mov<0xB1, 0xf>(p0 ? a2 : a3);  // pair sum of value 4 m + 2 + p0
        T q = (p1 ? w1 : w0) + dpp_mov<0x4E, 0xf>(p1 ? w0 : w1);         // quad sum of value 4 m + (lane & 3)
        q += dpp_mov<0x128, 0xf>(q);  // row_ror:8
        q += dpp_mov<0x124, 0xf>(q);  // row_ror:4 -> sum over the four quads of the row
        x[m] = q;
    }
#pragma unroll
    for (int j = 0; j < NV % 4; ++j) {
        T q = v[4 * (NV / 4) + j];
        q += dpp_mov<0xB1, 0xf>(q);
        q += dpp_mov<0x4E, 0xf>(q);
        q += dpp_mov<0x128, 0xf>(q);
        q += dpp_mov<0x124, 0xf>(q);
        y[j] = q;
    }
}

// Sum NV doubles across a workgroup of NW waves; every thread receives the totals in v[].
// scratch: NW*NV doubles of LDS.  Contains two barriers.
template <int NV, int NW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (NW == 1) return;
    __syncthreads();  // scratch may still be read by a previous round
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += scratch[w * NV + i];
        v[i] = s;
    }
}

// ---- 3x3 linear algebra in double (row-major) ---------------------------------------------------
__device__ __forceinline__ double det3(const double *a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
           a[2] * (a[3] * a[7] - a[4] * a[6]);
}

__device__ __forceinline__ void mul3(const double *a, const double *b, double *c) {
    double r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = r[i];
}

// One Jacobi rotation of columns (P,Q) of W (and V), compile-time column ids so everything stays in registers.
template <int P, int Q>
__device__ __forceinline__ bool jacobi_rotate(double *W, double *V) {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        alpha += W[3 * i + P] * W[3 * i + P];
        beta += W[3 * i + Q] * W[3 * i + Q];
        gamma += W[3 * i + P] * W[3 * i + Q];
    }
    const double lim = sqrt(alpha * beta);
    if (gamma == 0.0 || fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.220446049250313e-16 * 0.25 * lim) return false;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double wp = W[3 * i + P], wq = W[3 * i + Q];
        W[3 * i + P] = c * wp - s * wq;
        W[3 * i + Q] = s * wp + c * wq;
        const double vp = V[3 * i + P], vq = V[3 * i + Q];
        V[3 * i + P] = c * vp - s * vq;
        V[3 * i + Q] = s * vp + c * vq;
    }
    return true;
}

// Second half of the one-sided Jacobi SVD: W = A V has mutually orthogonal columns; sort them by norm
// (descending), normalise into U, complete the directions that carry no signal.
__device__ __forceinline__ void svd3_finish(double *W, double *Vm, double *U, double *S, double *V) {
    double s0 = sqrt(W[0] * W[0] + W[3] * W[3] + W[6] * W[6]);
    double s1 = sqrt(W[1] * W[1] + W[4] * W[4] + W[7] * W[7]);
    double s2 = sqrt(W[2] * W[2] + W[5] * W[5] + W[8] * W[8]);
    // sort columns descending with compile-time swaps (keeps arrays in registers)
#define F4L_SWAPCOL(a, b)                                                        \
    {                                                                            \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                          \
            double tw = W[3 * i + a]; W[3 * i + a] = W[3 * i + b]; W[3 * i + b] = tw; \
            double tv = Vm[3 * i + a]; Vm[3 * i + a] = Vm[3 * i + b]; Vm[3 * i + b] = tv; \
        }                                                                        \
    }
    if (s1 > s0) { double t = s0; s0 = s1; s1 = t; F4L_SWAPCOL(0, 1) }
    if (s2 > s0) { double t = s0; s0 = s2; s2 = t; F4L_SWAPCOL(0, 2) }
    if (s2 > s1) { double t = s1; s1 = s2; s2 = t; F4L_SWAPCOL(1, 2) }
#undef F4L_SWAPCOL
    S[0] = s0; S[1] = s1; S[2] = s2;
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = Vm[i];
    // U: first column by normalisation, second by Gram-Schmidt, third as a cross product.  Directions that
    // carry no signal (rank-deficient A) are completed from the matching columns of V, so that U V^T is the
    // identity on the null space (a zero matrix gives U = V, like a Jacobi SVD that never rotates).
    double u0[3], u1[3], u2[3];
    if (s0 > 0.0) {
        u0[0] = W[0] / s0; u0[1] = W[3] / s0; u0[2] = W[6] / s0;
    } else {
        u0[0] = Vm[0]; u0[1] = Vm[3]; u0[2] = Vm[6];
    }
    double pr = u0[0] * W[1] + u0[1] * W[4] + u0[2] * W[7];
    u1[0] = W[1] - pr * u0[0]; u1[1] = W[4] - pr * u0[1]; u1[2] = W[7] - pr * u0[2];
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (!(n1 > 1e-14 * s0 && n1 > 0.0)) {
        pr = u0[0] * Vm[1] + u0[1] * Vm[4] + u0[2] * Vm[7];
        u1[0] = Vm[1] - pr * u0[0]; u1[1] = Vm[4] - pr * u0[1]; u1[2] = Vm[7] - pr * u0[2];
        n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        if (!(n1 > 1e-8)) {
            const double ax = fabs(u0[0]), ay = fabs(u0[1]), az = fabs(u0[2]);
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            if (ax <= ay) { if (ax <= az) e0 = 1.0; else e2 = 1.0; }
            else { if (ay <= az) e1 = 1.0; else e2 = 1.0; }
            u1[0] = u0[1] * e2 - u0[2] * e1;
            u1[1] = u0[2] * e0 - u0[0] * e2;
            u1[2] = u0[0] * e1 - u0[1] * e0;
            n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        }
    }
    u1[0] /= n1; u1[1] /= n1; u1[2] /= n1;
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    {
        const bool sig = s2 > 1e-14 * s0 && s2 > 0.0;
        const double r0 = sig ? W[2] : Vm[2], r1 = sig ? W[5] : Vm[5], r2 = sig ? W[8] : Vm[8];
        if (u2[0] * r0 + u2[1] * r1 + u2[2] * r2 < 0.0) { u2[0] = -u2[0]; u2[1] = -u2[1]; u2[2] = -u2[2]; }
    }
    U[0] = u0[0]; U[1] = u1[0]; U[2] = u2[0];
    U[3] = u0[1]; U[4] = u1[1]; U[5] = u2[1];
    U[6] = u0[2]; U[7] = u1[2]; U[8] = u2[2];
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: A = U diag(S) V^T, S descending, U and V orthogonal also for
// rank-deficient A (free directions completed by Gram-Schmidt / cross product).
__device__ __forceinline__ void svd3(const double *A, double *U, double *S, double *V) {
    double W[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int i = 0; i < 9; ++i) W[i] = A[i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool any = false;
        any |= jacobi_rotate<0, 1>(W, Vm);
        any |= jacobi_rotate<0, 2>(W, Vm);
        any |= jacobi_rotate<1, 2>(W, Vm);
        if (!any) break;
    }
    svd3_finish(W, Vm, U, S, V);
}

// ---- double precision reciprocal / reciprocal square root without the IEEE fix-up sequences -----------------
// v_rcp_f64 / v_rsq_f64 deliver ~1e-7 relative; two Newton steps bring them to a couple of ulps (not correctly
// rounded, no denormal / inf handling): ~8 dependent instructions instead of ~25.  For normal, positive x.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * __builtin_fma(-x, y, 2.0);
    y = y * __builtin_fma(-x, y, 2.0);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    return y;
}
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0.0 ? x * fast_rsqrt(x) : 0.0; }

// svd3_finish for the ICP solve: same decisions, squared norms instead of norms (no square roots before the
// sort), normalisation by fast_rsqrt; the singular values themselves are not needed there.
__device__ __forceinline__ void svd3_finish_fast(double *W, double *Vm, double *U, double *V) {
    double q0 = W[0] * W[0] + W[3] * W[3] + W[6] * W[6];
    double q1 = W[1] * W[1] + W[4] * W[4] + W[7] * W[7];
    double q2 = W[2] * W[2] + W[5] * W[5] + W[8] * W[8];
#define F4L_SWAPCOL(a, b)                                                        \
    {                                                                            \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                          \
            double tw = W[3 * i + a]; W[3 * i + a] = W[3 * i + b]; W[3 * i + b] = tw; \
            double tv = Vm[3 * i + a]; Vm[3 * i + a] = Vm[3 * i + b]; Vm[3 * i + b] = tv; \
        }                                                                        \
    }
    if (q1 > q0) { double t = q0; q0 = q1; q1 = t; F4L_SWAPCOL(0, 1) }
    if (q2 > q0) { double t = q0; q0 = q2; q2 = t; F4L_SWAPCOL(0, 2) }
    if (q2 > q1) { double t = q1; q1 = q2; q2 = t; F4L_SWAPCOL(1, 2) }
#undef F4L_SWAPCOL
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = Vm[i];
    double u0[3], u1[3], u2[3];
    if (q0 > 1e-290) {
        const double inv = fast_rsqrt(q0);
        u0[0] = W[0] * inv; u0[1] = W[3] * inv; u0[2] = W[6] * inv;
    } else {
        u0[0] = Vm[0]; u0[1] = Vm[3]; u0[2] = Vm[6];
    }
    double pr = u0[0] * W[1] + u0[1] * W[4] + u0[2] * W[7];
    u1[0] = W[1] - pr * u0[0]; u1[1] = W[4] - pr * u0[1]; u1[2] = W[7] - pr * u0[2];
    double m1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
    if (!(m1 > 1e-28 * q0 && m1 > 1e-290)) {  // second direction carries no signal: complete it from V
        pr = u0[0] * Vm[1] + u0[1] * Vm[4] + u0[2] * Vm[7];
        u1[0] = Vm[1] - pr * u0[0]; u1[1] = Vm[4] - pr * u0[1]; u1[2] = Vm[7] - pr * u0[2];
        m1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
        if (!(m1 > 1e-16)) {
            const double ax = fabs(u0[0]), ay = fabs(u0[1]), az = fabs(u0[2]);
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            if (ax <= ay) { if (ax <= az) e0 = 1.0; else e2 = 1.0; }
            else { if (ay <= az) e1 = 1.0; else e2 = 1.0; }
            u1[0] = u0[1] * e2 - u0[2] * e1;
            u1[1] = u0[2] * e0 - u0[0] * e2;
            u1[2] = u0[0] * e1 - u0[1] * e0;
            m1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
        }
    }
    {
        const double inv = fast_rsqrt(m1);
        u1[0] *= inv; u1[1] *= inv; u1[2] *= inv;
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    {
        const bool sig = q2 > 1e-28 * q0 && q2 > 0.0;
        const double r0 = sig ? W[2] : Vm[2], r1 = sig ? W[5] : Vm[5], r2 = sig ? W[8] : Vm[8];
        if (u2[0] * r0 + u2[1] * r1 + u2[2] * r2 < 0.0) { u2[0] = -u2[0]; u2[1] = -u2[1]; u2[2] = -u2[2]; }
    }
    U[0] = u0[0]; U[1] = u1[0]; U[2] = u2[0];
    U[3] = u0[1]; U[4] = u1[1]; U[5] = u2[1];
    U[6] = u0[2]; U[7] = u1[2]; U[8] = u2[2];
}

// ---- the same SVD for the per-iteration solve of the ICP kernel, built for latency ------------------------
// A rotation only has to (a) be orthogonal to working precision and (b) shrink the off-diagonal term; how
// accurately its ANGLE is computed merely steers convergence.  So the angle uses the raw v_rcp_f64 /
// v_sqrt_f64 estimates (~1e-7 relative, one instruction each instead of a ~12-instruction IEEE sequence), and
// only c = (1 + t^2)^(-1/2) is polished by two Newton steps to full double precision; s = c t then makes
// c^2 + s^2 = 1 to rounding.  Columns count as orthogonal at |gamma| <= 8 eps sqrt(alpha beta): the exit
// test of svd3 (eps / 4) sits below the rounding noise of gamma and usually burns all 60 sweeps.
// Returns gamma^2 / (alpha beta) BEFORE the rotation (0 when the pair already counts as orthogonal).
template <int P, int Q>
__device__ __forceinline__ double jacobi_rotate_fast(double *W, double *V) {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        alpha += W[3 * i + P] * W[3 * i + P];
        beta += W[3 * i + Q] * W[3 * i + Q];
        gamma += W[3 * i + P] * W[3 * i + Q];
    }
    const double tol = 8.0 * 2.220446049250313e-16;
    const double g2 = gamma * gamma, ab = alpha * beta;
    if (!(g2 > (tol * tol) * ab)) return 0.0;  // converged pair (also gamma == 0, NaN)
    const double zeta = (beta - alpha) * 0.5 * __builtin_amdgcn_rcp(gamma);
    double t = __builtin_amdgcn_rcp(fabs(zeta) + __builtin_amdgcn_sqrt(__builtin_fma(zeta, zeta, 1.0)));
    t = zeta < 0.0 ? -t : t;
    const double x = __builtin_fma(t, t, 1.0);
    double c = __builtin_amdgcn_rsq(x);
    c = c * __builtin_fma(-0.5 * x * c, c, 1.5);
    c = c * __builtin_fma(-0.5 * x * c, c, 1.5);
    const double sn = c * t;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double wp = W[3 * i + P], wq = W[3 * i + Q];
        W[3 * i + P] = c * wp - sn * wq;
        W[3 * i + Q] = sn * wp + c * wq;
        const double vp = V[3 * i + P], vq = V[3 * i + Q];
        V[3 * i + P] = c * vp - sn * vq;
        V[3 * i + Q] = sn * vp + c * vq;
    }
    return g2 * __builtin_amdgcn_rcp(ab);
}

// A = U diag(S) V^T with the sweeps WARM-STARTED from V0 (orthogonal; the V of the previous ICP iteration, or I):
// W = A V0 already has nearly orthogonal columns when A moved little, so one or two sweeps finish it.
__device__ __forceinline__ int svd3_warm(const double *A, const double *V0, double *U, double *V) {
    double W[9], Vm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Vm[i] = V0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            W[3 * i + j] = A[3 * i] * V0[j] + A[3 * i + 1] * V0[3 + j] + A[3 * i + 2] * V0[6 + j];
    int sweeps = 0;
#pragma nounroll
    for (; sweeps < 16; ++sweeps) {
        double worst = jacobi_rotate_fast<0, 1>(W, Vm);
        worst = fmax(worst, jacobi_rotate_fast<0, 2>(W, Vm));
        worst = fmax(worst, jacobi_rotate_fast<1, 2>(W, Vm));
        // Jacobi converges quadratically: relative off-diagonals below 1e-8 before a sweep are below the
        // threshold after it, so no verification sweep is needed (the angle's 1e-7 error leaves 1e-15 at most)
        if (worst < 1e-16) break;
    }
    svd3_finish_fast(W, Vm, U, V);
    return sweeps;
}

// ---- optimal rotation by Newton's method on SO(3) ---------------------------------------------------------
// R maximising tr(R^T S) over rotations (S = cross covariance, rows: target, columns: source) is what Kabsch /
// Umeyama read off the SVD of S (U diag(1,1,det) V^T).  ICP asks for it once per iteration with clouds that are
// already nearly aligned, so S is close to symmetric positive semi-definite and the answer close to I: Newton
// from R = I converges quadratically.  With B = R^T S:  gradient  v = (B21 - B12, B02 - B20, B10 - B01),
// Hessian  G = tr(B) I - sym(B)  (eigenvalues: pairwise sums of B's singular values when B is symmetric),
// step  w = G^-1 v,  R <- R cayley(w / 2)  (exactly orthogonal whatever w is).  ~140 flops per step, 2-3 steps,
// instead of ~10 Jacobi rotations plus the U/V clean-up.  Returns false -- caller falls back to the SVD -- when
// G is not safely positive definite (rank-deficient or badly misaligned input) or the steps do not shrink.
#ifndef F4L_NEWTON_DONE
#define F4L_NEWTON_DONE 1e-15
#endif
__device__ __forceinline__ bool rot_newton(const double *S, double *R) {
    double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, B[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) B[i] = S[i];
    bool ok = true, done = false;
#pragma nounroll
    for (int it = 0; it < 6 && ok && !done; ++it) {
        const double v0 = B[7] - B[5], v1 = B[2] - B[6], v2 = B[3] - B[1];
        const double tr = B[0] + B[4] + B[8];
        const double g00 = tr - B[0], g11 = tr - B[4], g22 = tr - B[8];
        const double g01 = -0.5 * (B[1] + B[3]), g02 = -0.5 * (B[2] + B[6]), g12 = -0.5 * (B[5] + B[7]);
        // cofactors of the symmetric G; positive definiteness from the leading minors, scaled by tr(G) = 2 tr(B)
        const double c00 = g11 * g22 - g12 * g12, c01 = g02 * g12 - g01 * g22, c02 = g01 * g12 - g02 * g11;
        const double c11 = g00 * g22 - g02 * g02, c12 = g01 * g02 - g00 * g12, c22 = g00 * g11 - g01 * g01;
        const double det = g00 * c00 + g01 * c01 + g02 * c02;
        const double sc = 2.0 * tr;
        ok = sc > 0.0 && g00 > 1e-9 * sc && c22 > 1e-9 * sc * sc && det > 1e-9 * sc * sc * sc && det < 1e300;
        if (!ok) break;
        const double id = fast_rcp(det);
        const double w0 = (c00 * v0 + c01 * v1 + c02 * v2) * id;
        const double w1 = (c01 * v0 + c11 * v1 + c12 * v2) * id;
        const double w2 = (c02 * v0 + c12 * v1 + c22 * v2) * id;
        const double ww = w0 * w0 + w1 * w1 + w2 * w2;
        ok = ww < 1.0;                 // a step beyond ~1 rad: not the regime this is meant for
        // |w| < 3e-8: what is left after this step is ~|w|^2.  (Round 4 measured |w| < 1e-6 -- one Newton step fewer in the nearly
        // converged passes, 1e-12 rad left -- and found NOTHING: C4 18.16 ms either way, the supervoxel tile 1.23 ms either way.
        // The solve is a chain the patch waits for, but its length is not in the Newton steps.)
        done = ww < F4L_NEWTON_DONE;
        // C = I + 2 / (1 + |u|^2) ([u]x + [u]x^2),  u = w / 2
        const double u0 = 0.5 * w0, u1 = 0.5 * w1, u2 = 0.5 * w2;
        const double f = 2.0 * fast_rcp(1.0 + 0.25 * ww);
        double C[9];
        C[0] = 1.0 - f * (u1 * u1 + u2 * u2); C[1] = f * (u0 * u1 - u2);       C[2] = f * (u0 * u2 + u1);
        C[3] = f * (u0 * u1 + u2);       C[4] = 1.0 - f * (u0 * u0 + u2 * u2); C[5] = f * (u1 * u2 - u0);
        C[6] = f * (u0 * u2 - u1);       C[7] = f * (u1 * u2 + u0);       C[8] = 1.0 - f * (u0 * u0 + u1 * u1);
        mul3(Rm, C, Rm);  // R <- R C
        // B <- C^T B
        double Bn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Bn[3 * i + j] = C[i] * B[j] + C[3 + i] * B[3 + j] + C[6 + i] * B[6 + j];
#pragma unroll
        for (int i = 0; i < 9; ++i) B[i] = Bn[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rm[i];
    return ok && done;
}

// R = A diag(1,1,d) B^T for row-major 3x3 A, B.
__device__ __forceinline__ void mul_diag_bt(const double *A, double d, const double *B, double *R) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + d * A[3 * i + 2] * B[3 * j + 2];
}

}  // namespace f4l
