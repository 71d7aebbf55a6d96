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
#include "topk.h"

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

// One workgroup per patch, one wave per query point (queries strided over the 4 waves).
__global__ __launch_bounds__(PN_NT) void patch_normals_kernel(const float *__restrict__ pts,
                                                               const int64_t *__restrict__ off, int64_t P, int knn,
                                                               int lds_cap, float *__restrict__ normals) {
    extern __shared__ __attribute__((aligned(16))) float pl[];  // packed xyz of the patch
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = off[p];
    const int n = (int)(off[p + 1] - o);
    if (n == 0) return;
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
            best.offer(cd, ci, k);
        }
        // cumulants of the k neighbours (self included), Open3D ComputeCovariance
        double cum[9];
        {
            const bool have = lane < k;
            const int j = have ? best.i : q;
            const double x = have ? (double)base[3 * j] : 0.0, y = have ? (double)base[3 * j + 1] : 0.0,
                         z = have ? (double)base[3 * j + 2] : 0.0;
            cum[0] = x; cum[1] = y; cum[2] = z;
            cum[3] = x * x; cum[4] = x * y; cum[5] = x * z; cum[6] = y * y; cum[7] = y * z; cum[8] = z * z;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) cum[i] = wave_sum(cum[i]);
        double nv[3];
        if (k < 3) {
            nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0;  // identity covariance -> no preferred axis -> (0,0,1)
        } else {
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
        if (lane == 0) {
            float *out = normals + 3 * (o + q);
            out[0] = (float)nv[0]; out[1] = (float)nv[1]; out[2] = (float)nv[2];
        }
    }
}

// refine_dvfs_with_threshold: one workgroup per patch; target patch in LDS as float4 relative to its first
// point; every lane owns source points, brute-force nearest neighbour, validity d2 < thr^2.
__global__ __launch_bounds__(PN_NT) void nn_refine_kernel(const float *__restrict__ src, const int64_t *__restrict__ src_off,
                                                           const float *__restrict__ tgt, const int64_t *__restrict__ tgt_off,
                                                           int64_t P, const double *__restrict__ T,
                                                           const double *__restrict__ thr, int lds_cap,
                                                           int32_t *__restrict__ nn_out, float *__restrict__ out6) {
    extern __shared__ __attribute__((aligned(16))) float4 tl4[];
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t s0 = src_off[p], t0 = tgt_off[p];
    const int ns = (int)(src_off[p + 1] - s0), nt = (int)(tgt_off[p + 1] - t0);
    const float *__restrict__ sg = src + 3 * s0;
    const float *__restrict__ tg = tgt + 3 * t0;
    const int tid = (int)threadIdx.x;
    const bool in_lds = nt <= lds_cap;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    if (in_lds) {
        for (int j = tid; j < nt; j += PN_NT) tl4[j] = make_float4(tg[3 * j] - ox, tg[3 * j + 1] - oy, tg[3 * j + 2] - oz, 0.f);
        __syncthreads();
    }
    const double *Tp = T + 16 * p;
    // p' = R (s' + o) + t - o with s' = s - o
    float Rf[9], tf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Rf[3 * i] = (float)Tp[4 * i]; Rf[3 * i + 1] = (float)Tp[4 * i + 1]; Rf[3 * i + 2] = (float)Tp[4 * i + 2];
        const double o_i = i == 0 ? ox : (i == 1 ? oy : oz);
        tf[i] = (float)(Tp[4 * i] * (double)ox + Tp[4 * i + 1] * (double)oy + Tp[4 * i + 2] * (double)oz + Tp[4 * i + 3] - o_i);
    }
    const double th = thr[p];
    const float th2 = (float)(th * th);
    for (int i = tid; i < ns; i += PN_NT) {
        const float sx = sg[3 * i], sy = sg[3 * i + 1], sz = sg[3 * i + 2];
        const float x = sx - ox, y = sy - oy, z = sz - oz;
        const float px = Rf[0] * x + Rf[1] * y + Rf[2] * z + tf[0];
        const float py = Rf[3] * x + Rf[4] * y + Rf[5] * z + tf[1];
        const float pz = Rf[6] * x + Rf[7] * y + Rf[8] * z + tf[2];
        float best = __builtin_inff();
        int bj = -1;
        if (in_lds) {
#pragma unroll 8
            for (int j = 0; j < nt; ++j) {
                const float4 q = tl4[j];
                const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; bj = j; }
            }
        } else {
            for (int j = 0; j < nt; ++j) {
                const float dx = px - (tg[3 * j] - ox), dy = py - (tg[3 * j + 1] - oy), dz = pz - (tg[3 * j + 2] - oz);
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; bj = j; }
            }
        }
        const bool hit = bj >= 0 && best < th2;  // :80 dists[0] < distance_threshold ** 2
        if (nn_out) nn_out[s0 + i] = hit ? bj : -1;
        if (out6) {
            float *o6 = out6 + 6 * (s0 + i);
            o6[0] = sx; o6[1] = sy; o6[2] = sz;
            o6[3] = hit ? tg[3 * bj] : 0.f; o6[4] = hit ? tg[3 * bj + 1] : 0.f; o6[5] = hit ? tg[3 * bj + 2] : 0.f;
        }
    }
}

}  // namespace f4l

extern "C" int f4l_patch_normals(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                                 float *normals_out, void *stream) {
    using namespace f4l;
    if (P < 0 || !off || knn < 1 || max_patch_host < 0 || (max_patch_host > 0 && (!pts || !normals_out))) return F4L_EINVAL;
    if (knn > F4L_MAX_K) return F4L_EUNSUPPORTED;
    if (P == 0 || max_patch_host == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_patch_host > 0x3fffffffLL) return F4L_EUNSUPPORTED;
    const int cap = (int)(max_patch_host < PN_LDS_MAX ? max_patch_host : PN_LDS_MAX);
    const size_t lds = (size_t)cap * 12;
    if (lds > 64 * 1024)
        F4L_HIP_CHECK(hipFuncSetAttribute((const void *)patch_normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(patch_normals_kernel, dim3((unsigned)P), dim3(PN_NT), lds, (hipStream_t)stream, pts, off, P, knn,
                       cap, normals_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

extern "C" int f4l_nn_refine(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                             const double *T, const double *thr, int64_t max_tgt_patch_host, int32_t *nn_out, float *out6,
                             void *stream) {
    using namespace f4l;
    if (P < 0 || !src_off || !tgt_off || !T || !thr || max_tgt_patch_host < 0) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    if (P > 0x7fffffffLL || max_tgt_patch_host > 0x3fffffffLL) return F4L_EUNSUPPORTED;
    const int cap = (int)(max_tgt_patch_host < 8192 ? max_tgt_patch_host : 8192);
    const size_t lds = (size_t)cap * 16;
    if (lds > 64 * 1024)
        F4L_HIP_CHECK(hipFuncSetAttribute((const void *)nn_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nn_refine_kernel, dim3((unsigned)P), dim3(PN_NT), lds, (hipStream_t)stream, src, src_off, tgt,
                       tgt_off, P, T, thr, cap, nn_out, out6);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
