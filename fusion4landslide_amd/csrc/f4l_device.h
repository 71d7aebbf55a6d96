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

// R = A diag(1,1,d) B^T for row-major 3x3 A, B.
__device__ __forceinline__ void mul_diag_bt(const double *A, double d, const double *B, double *R) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + d * A[3 * i + 2] * B[3 * j + 2];
}

}  // namespace f4l
