// core.hip -- library plumbing: version, error strings, device query, elementwise helpers.
#include <string.h>

#include "f4l_device.h"

thread_local int f4l_tls_hip_error = 0;

extern "C" int f4l_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *f4l_strerror(int code) {
    switch (code) {
        case F4L_OK: return "ok";
        case F4L_EINVAL: return "invalid argument";
        case F4L_EWORKSPACE: return "workspace too small";
        case F4L_EHIP: return "HIP runtime error (see f4l_last_hip_error)";
        case F4L_EUNSUPPORTED: return "unsupported request";
        case F4L_ENOMEM: return "host allocation failed";
        default: return "unknown error";
    }
}

extern "C" int f4l_last_hip_error(void) { return f4l_tls_hip_error; }

extern "C" int f4l_device_info(int *cu_count, int *lds_bytes, int64_t *hbm_bytes, char *arch, int arch_len) {
    int dev = 0;
    F4L_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    F4L_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return F4L_OK;
}

namespace f4l {

// a14: rows [s, T_p s] (or [T_p^-1 q, q]) for every point; one workgroup walks one patch so that the
// 4x4 is read once per workgroup.  12 B read + 24 B written per point: pure HBM streaming.
__global__ void apply_transform_kernel(const float *__restrict__ pts, const int64_t *__restrict__ off, int64_t P,
                                       const double *__restrict__ T, int inverse, float *__restrict__ out6) {
    for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
        const int64_t o = off[p];
        const int n = (int)(off[p + 1] - o);
        const double *t = T + 16 * p;
        double r[9], tr[3];
        if (!inverse) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r[3 * i] = t[4 * i]; r[3 * i + 1] = t[4 * i + 1]; r[3 * i + 2] = t[4 * i + 2];
                tr[i] = t[4 * i + 3];
            }
        } else {  // x = R^T (q - t)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r[3 * i] = t[i]; r[3 * i + 1] = t[4 + i]; r[3 * i + 2] = t[8 + i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) tr[i] = -(r[3 * i] * t[3] + r[3 * i + 1] * t[7] + r[3 * i + 2] * t[11]);
        }
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
            const float xf = pts[3 * (o + i)], yf = pts[3 * (o + i) + 1], zf = pts[3 * (o + i) + 2];
            const double x = xf, y = yf, z = zf;
            const float ax = (float)(r[0] * x + r[1] * y + r[2] * z + tr[0]);
            const float ay = (float)(r[3] * x + r[4] * y + r[5] * z + tr[1]);
            const float az = (float)(r[6] * x + r[7] * y + r[8] * z + tr[2]);
            float *o6 = out6 + 6 * (o + i);
            if (!inverse) { o6[0] = xf; o6[1] = yf; o6[2] = zf; o6[3] = ax; o6[4] = ay; o6[5] = az; }
            else { o6[0] = ax; o6[1] = ay; o6[2] = az; o6[3] = xf; o6[4] = yf; o6[5] = zf; }
        }
    }
}

__global__ void gather_points_kernel(const float *__restrict__ pts, const int32_t *__restrict__ order, int64_t n,
                                     float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = order[i];
        out[3 * i] = pts[3 * j]; out[3 * i + 1] = pts[3 * j + 1]; out[3 * i + 2] = pts[3 * j + 2];
    }
}

}  // namespace f4l

extern "C" int f4l_apply_transform(const float *pts, const int64_t *off, int64_t P, int64_t n_total, const double *T,
                                   int inverse, float *out6, void *stream) {
    if (P < 0 || n_total < 0 || !off || !T || (n_total > 0 && (!pts || !out6))) return F4L_EINVAL;
    if (P == 0 || n_total == 0) return F4L_OK;
    const unsigned grid = (unsigned)(P < 65536 ? P : 65536);
    hipLaunchKernelGGL(f4l::apply_transform_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pts, off, P, T,
                       inverse, out6);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

extern "C" int f4l_gather_points(const float *pts, const int32_t *order, int64_t n, float *out, void *stream) {
    if (n < 0 || (n > 0 && (!pts || !order || !out))) return F4L_EINVAL;
    if (n == 0) return F4L_OK;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(f4l::gather_points_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, pts, order, n, out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
