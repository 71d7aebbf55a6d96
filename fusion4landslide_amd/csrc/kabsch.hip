// kabsch.hip -- batched ragged weighted Kabsch / Procrustes for gfx950.
//
// Replaces scripts/weighted_svd.py:58-129 (`weighted_procrustes`) as called once per patch at
// src/coarse_to_fine_matching_base.py:3341.  The reference issues ~12 tiny torch kernels plus torch.svd
// per patch (B = 1); here one workgroup handles one patch and the whole batch is one launch.
//
// Data flow per patch (n rows): two streaming passes over [s_i, r_i, w_i] (28 B/row, second pass served
// from L1/L2), all accumulation in double, wave __shfl reductions, 3x3 one-sided Jacobi SVD on the
// totals.  HBM-bound by construction (contraction is 3x3: no MFMA).  Algorithmic bytes: 24 B/row (+4 with
// weights) read, 96 B/patch written.
#include "f4l_device.h"

namespace f4l {

// V2 = false: scripts/weighted_svd.py:58-129 (weighted_procrustes).  V2 = true: src/functions.py:12-85
// (kabsch_transformation_estimation, the F2S3 variant): weights normalised by (sum w + eps) first (when
// normalize_w), thresholded AFTER that and only when the threshold is positive, means divided by (sum w' + eps)
// again, covariance with the normalised weights, and the third column scaled by det(V U^T) itself, not its sign.
template <typename T, int NW, bool V2>
__global__ __launch_bounds__(NW * 64) void kabsch_kernel(const T *__restrict__ src, const T *__restrict__ ref,
                                                         const T *__restrict__ w, const int64_t *__restrict__ off,
                                                         int64_t P, double w_thresh, double eps, int normalize_w,
                                                         double *__restrict__ R_out, double *__restrict__ t_out,
                                                         double *__restrict__ T_out) {
    __shared__ double scratch[NW * 9];
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = off[p];
    const int n = (int)(off[p + 1] - o);
    const T *s = src + 3 * o, *r = ref + 3 * o;
    const T *ww = w ? w + o : nullptr;
    const int tid = (int)threadIdx.x, NT = NW * 64;

    double pre = 1.0;  // V2: 1 / (sum w + eps) applied to every weight before anything else (functions.py:35-37)
    if (V2 && normalize_w) {
        double sw[1] = {0.0};
        for (int i = tid; i < n; i += NT) sw[0] += ww ? (double)ww[i] : 1.0;
        block_sum<1, NW>(sw, scratch);
        pre = 1.0 / (sw[0] + eps);
        __syncthreads();
    }
    // pass 1: sum w, sum w s, sum w r   (weighted_svd.py:94-100)
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += NT) {
        double wi = ww ? (double)ww[i] : 1.0;
        if (V2) { wi *= pre; if (w_thresh > 0.0 && wi < w_thresh) wi = 0.0; }
        else if (wi < w_thresh) wi = 0.0;
        a[0] += wi;
        a[1] += wi * (double)s[3 * i];
        a[2] += wi * (double)s[3 * i + 1];
        a[3] += wi * (double)s[3 * i + 2];
        a[4] += wi * (double)r[3 * i];
        a[5] += wi * (double)r[3 * i + 1];
        a[6] += wi * (double)r[3 * i + 2];
    }
    block_sum<7, NW>(a, scratch);
    const double inv = 1.0 / (a[0] + eps);  // eps stays in the denominator (:96)
    const double cs0 = a[1] * inv, cs1 = a[2] * inv, cs2 = a[3] * inv;
    const double ct0 = a[4] * inv, ct1 = a[5] * inv, ct2 = a[6] * inv;

    // pass 2: H = sum (s - cs) w' (r - ct)^T   (:101-104)
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += NT) {
        double wi = ww ? (double)ww[i] : 1.0;
        if (V2) { wi *= pre; if (w_thresh > 0.0 && wi < w_thresh) wi = 0.0; }
        else { if (wi < w_thresh) wi = 0.0; wi *= inv; }
        const double a0 = (double)s[3 * i] - cs0, a1 = (double)s[3 * i + 1] - cs1, a2 = (double)s[3 * i + 2] - cs2;
        const double b0 = wi * ((double)r[3 * i] - ct0), b1 = wi * ((double)r[3 * i + 1] - ct1),
                     b2 = wi * ((double)r[3 * i + 2] - ct2);
        h[0] += a0 * b0; h[1] += a0 * b1; h[2] += a0 * b2;
        h[3] += a1 * b0; h[4] += a1 * b1; h[5] += a1 * b2;
        h[6] += a2 * b0; h[7] += a2 * b1; h[8] += a2 * b2;
    }
    block_sum<9, NW>(h, scratch);

    if (tid == 0) {
        double U[9], V[9], R[9];
        if (n == 0) {
            R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        } else {
            // Newton on SO(3) first (f4l_device.h: the maximiser of tr(R^T H^T), which is V diag(1,1,sign det) U^T whenever
            // that is unique), the Jacobi SVD when the problem is rank deficient, near a reflection tie, or the
            // rotation is large
            const double Ht[9] = {h[0], h[3], h[6], h[1], h[4], h[7], h[2], h[5], h[8]};
            if (!rot_newton(Ht, R)) {
                const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                svd3_warm(h, I3, U, V);  // the latency-trimmed Jacobi of f4l_device.h (cold start)
                const double d = det3(V) * det3(U);  // det(V U^T)
                const double sg = V2 ? d : (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0));  // torch.sign (:111) / the determinant itself (functions.py:70-72)
                mul_diag_bt(V, sg, U, R);
            }
        }
        const double t0 = ct0 - (R[0] * cs0 + R[1] * cs1 + R[2] * cs2);  // :113
        const double t1 = ct1 - (R[3] * cs0 + R[4] * cs1 + R[5] * cs2);
        const double t2 = ct2 - (R[6] * cs0 + R[7] * cs1 + R[8] * cs2);
        if (R_out) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R_out[9 * p + i] = R[i];
        }
        if (t_out) { t_out[3 * p + 0] = t0; t_out[3 * p + 1] = t1; t_out[3 * p + 2] = t2; }
        if (T_out) {  // return_transform=True: the 4x4 of scripts/weighted_svd.py:115-120
            double *M = T_out + 16 * p;
            M[0] = R[0]; M[1] = R[1]; M[2] = R[2]; M[3] = t0;
            M[4] = R[3]; M[5] = R[4]; M[6] = R[5]; M[7] = t1;
            M[8] = R[6]; M[9] = R[7]; M[10] = R[8]; M[11] = t2;
            M[12] = 0.0; M[13] = 0.0; M[14] = 0.0; M[15] = 1.0;
        }
    }
}

// residual norms per row, scripts/weighted_svd.py:143-146
__global__ void kabsch_residual_kernel(const float *__restrict__ src, const float *__restrict__ ref,
                                       const int64_t *__restrict__ off, int64_t P, const double *__restrict__ R,
                                       const double *__restrict__ t, double *__restrict__ res) {
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int64_t o = off[p];
    const int n = (int)(off[p + 1] - o);
    double r[9], tt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = R[9 * p + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = t[3 * p + i];
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
        const double x = src[3 * (o + i)], y = src[3 * (o + i) + 1], z = src[3 * (o + i) + 2];
        const double dx = r[0] * x + r[1] * y + r[2] * z + tt[0] - (double)ref[3 * (o + i)];
        const double dy = r[3] * x + r[4] * y + r[5] * z + tt[1] - (double)ref[3 * (o + i) + 1];
        const double dz = r[6] * x + r[7] * y + r[8] * z + tt[2] - (double)ref[3 * (o + i) + 2];
        res[o + i] = sqrt(dx * dx + dy * dy + dz * dz);
    }
}

template <typename T, bool V2 = false>
static int launch_kabsch(const T *src, const T *ref, const T *w, const int64_t *off, int64_t P, int64_t n_total,
                         double w_thresh, double eps, double *R_out, double *t_out, double *T_out, hipStream_t st,
                         int normalize_w = 0) {
    if (P < 0 || n_total < 0 || !off || (!T_out && (!R_out || !t_out)) || (n_total > 0 && (!src || !ref))) return F4L_EINVAL;
    if (P == 0) return F4L_OK;
    if (P > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    // one wave per patch while patches are small (no LDS round, no barrier); four waves otherwise
    if (n_total / P <= 256)
        hipLaunchKernelGGL((kabsch_kernel<T, 1, V2>), dim3((unsigned)P), dim3(64), 0, st, src, ref, w, off, P, w_thresh,
                           eps, normalize_w, R_out, t_out, T_out);
    else
        hipLaunchKernelGGL((kabsch_kernel<T, 4, V2>), dim3((unsigned)P), dim3(256), 0, st, src, ref, w, off, P, w_thresh,
                           eps, normalize_w, R_out, t_out, T_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

}  // namespace f4l

extern "C" int f4l_kabsch_batched(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                                  int64_t n_total, double w_thresh, double eps, double *R_out, double *t_out,
                                  void *stream) {
    return f4l::launch_kabsch<float>(src, ref, w, off, P, n_total, w_thresh, eps, R_out, t_out, nullptr, (hipStream_t)stream);
}

extern "C" int f4l_kabsch_batched_f64(const double *src, const double *ref, const double *w, const int64_t *off,
                                      int64_t P, int64_t n_total, double w_thresh, double eps, double *R_out,
                                      double *t_out, void *stream) {
    return f4l::launch_kabsch<double>(src, ref, w, off, P, n_total, w_thresh, eps, R_out, t_out, nullptr, (hipStream_t)stream);
}

extern "C" int f4l_kabsch_transforms(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                                     int64_t n_total, double w_thresh, double eps, double *T_out, void *stream) {
    return f4l::launch_kabsch<float>(src, ref, w, off, P, n_total, w_thresh, eps, nullptr, nullptr, T_out, (hipStream_t)stream);
}

extern "C" int f4l_kabsch2_batched(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                                   int64_t n_total, int normalize_w, double w_thresh, double eps, double *R_out,
                                   double *t_out, void *stream) {
    return f4l::launch_kabsch<float, true>(src, ref, w, off, P, n_total, w_thresh, eps, R_out, t_out, nullptr,
                                           (hipStream_t)stream, normalize_w);
}

extern "C" int f4l_kabsch2_batched_f64(const double *src, const double *ref, const double *w, const int64_t *off,
                                       int64_t P, int64_t n_total, int normalize_w, double w_thresh, double eps,
                                       double *R_out, double *t_out, void *stream) {
    return f4l::launch_kabsch<double, true>(src, ref, w, off, P, n_total, w_thresh, eps, R_out, t_out, nullptr,
                                            (hipStream_t)stream, normalize_w);
}

extern "C" int f4l_kabsch_residuals(const float *src, const float *ref, const int64_t *off, int64_t P,
                                    int64_t n_total, const double *R, const double *t, double *res_out, void *stream) {
    if (P < 0 || n_total < 0 || !off || !R || !t || (n_total > 0 && (!src || !ref || !res_out))) return F4L_EINVAL;
    if (P == 0 || n_total == 0) return F4L_OK;
    if (P > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipLaunchKernelGGL(f4l::kabsch_residual_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, src, ref, off,
                       P, R, t, res_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
