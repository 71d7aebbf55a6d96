// supervoxel.hip -- the partition entry point: device kNN + device normals + host segmentation.
//
// Replaces `computeSupervoxel` (cpp_core/supervoxel_segmentation/supervoxel.cpp:83-143) minus file I/O.
// Stage split and why:
//   kNN (supervoxel.cpp:105-107)  and PCA normals (:108-113)  -> gfx950 kernels in knn.hip  (47 % of the
//        reference's time at 1 M points, embarrassingly parallel);
//   boundary-preserving segmentation (codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-265)
//        -> host, below.  Its fusion pass is sequential and ORDER DEPENDENT (the visiting order of the
//        representatives, a mutable adjacency, an early `break` when the count hits K), so a label-identical
//        result requires replaying that order.  It is written here as a flat, allocation-free host routine
//        (pooled adjacency lists, SoA points) rather than the reference's Array<Array<int>> of heap vectors.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <random>
#include <vector>

#include "f4l_device.h"
#include "sv_metric.h"
#include "supervoxel_host.h"

namespace f4l {
#pragma clang fp contract(off)
__global__ void sv_min_metric_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm,
                                     const int32_t *__restrict__ knn, int64_t n, int k, double resolution,
                                     double *__restrict__ dis0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double best = DBL_MAX;
    for (int j = 0; j < k; ++j) {
        const int64_t q = knn[i * k + j];
        if (q != i) {
            const double m = sv_metric(xyz, nrm, i, q, resolution);
            best = m < best ? m : best;  // std::min(best, m)
        }
    }
    dis0[i] = best;
}
__global__ void sv_boundary_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm,
                                   const int32_t *__restrict__ knn, const int32_t *__restrict__ labels, int64_t n, int k,
                                   double resolution, uint8_t *__restrict__ flag, double *__restrict__ dis) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = labels[i];
    bool f = false;
    for (int j = 0; j < k; ++j) f = f || labels[knn[i * k + j]] != l;
    flag[i] = f ? 1 : 0;
    dis[i] = sv_metric(xyz, nrm, i, (int64_t)l, resolution);
}

struct BoundaryCtx {
    const float *xyz;
    const double *nrm;
    const int32_t *knn;
    int32_t *d_labels;
    uint8_t *d_flag;
    double *d_dis;
    int64_t n;
    int k;
    double resolution;
    hipStream_t st;
    int rc;
};
static void boundary_on_device(const int32_t *labels, uint8_t *flag, double *dis, void *vctx) {
    BoundaryCtx &c = *(BoundaryCtx *)vctx;
    const unsigned grid = (unsigned)((c.n + 255) / 256);
    c.rc = F4L_EHIP;
    if (hipMemcpyAsync(c.d_labels, labels, (size_t)c.n * 4, hipMemcpyHostToDevice, c.st) != hipSuccess) return;
    hipLaunchKernelGGL(sv_boundary_kernel, dim3(grid), dim3(256), 0, c.st, c.xyz, c.nrm, c.knn, c.d_labels, c.n, c.k,
                       c.resolution, c.d_flag, c.d_dis);
    if (hipGetLastError() != hipSuccess) return;
    if (hipMemcpyAsync(flag, c.d_flag, (size_t)c.n, hipMemcpyDeviceToHost, c.st) != hipSuccess) return;
    if (hipMemcpyAsync(dis, c.d_dis, (size_t)c.n * 8, hipMemcpyDeviceToHost, c.st) != hipSuccess) return;
    if (hipStreamSynchronize(c.st) != hipSuccess) return;
    c.rc = F4L_OK;
}
}  // namespace f4l

extern "C" size_t f4l_supervoxel_segment_exact_workspace_bytes(int64_t n, int k);
extern "C" int f4l_supervoxel_segment_exact(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k,
                                            double resolution, int32_t *labels_out, int32_t *n_supervoxels_host, int32_t *stats_host,
                                            void *workspace, size_t workspace_bytes, void *stream);
static size_t sv_shared_bytes(int64_t n, int k) {  // the kNN's workspace, reused by the device segmentation once the search is over
    const size_t a = f4l_knn_workspace_bytes(n, k), x = f4l_supervoxel_segment_exact_workspace_bytes(n, k);
    return a > x ? a : x;
}
extern "C" size_t f4l_supervoxel_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    // kNN workspace + (when the caller does not want the intermediates) room for idx and normals + the two sweeps of
    // the segmentation that run on the device (8 n of metric values, n of flags)
    const size_t a = sv_shared_bytes(n, k);
    const size_t idx = ((size_t)n * k * 4 + 255) / 256 * 256, nrm = ((size_t)n * 24 + 255) / 256 * 256;
    const size_t dis = ((size_t)n * 8 + 255) / 256 * 256, flag = ((size_t)n + 255) / 256 * 256;
    return a + idx + nrm + dis + flag;
}

// SYNCHRONISES `stream`: kNN and normals run on the device, the order-dependent segmentation on the host.
extern "C" int f4l_supervoxel(const float *xyz, int64_t n, int k, double resolution, int32_t *labels_out,
                              int32_t *n_supervoxels_host, int32_t *knn_out, double *normals_out, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!xyz || n <= 0 || k < 1 || k >= n || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;  // supervoxel.cpp:100
    if (k > F4L_MAX_K || n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    if (workspace_bytes < f4l_supervoxel_workspace_bytes(n, k)) return F4L_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t knn_ws = sv_shared_bytes(n, k);
    unsigned char *base = (unsigned char *)workspace;
    const size_t idx_b = ((size_t)n * k * 4 + 255) / 256 * 256, nrm_b = ((size_t)n * 24 + 255) / 256 * 256,
                 dis_b = ((size_t)n * 8 + 255) / 256 * 256;
    int32_t *idx = knn_out ? knn_out : (int32_t *)(base + knn_ws);
    double *nrm = normals_out ? normals_out : (double *)(base + knn_ws + idx_b);
    double *d_dis = (double *)(base + knn_ws + idx_b + nrm_b);
    uint8_t *d_flag = (uint8_t *)(base + knn_ws + idx_b + nrm_b + dis_b);
    int rc = f4l_knn_normals(xyz, n, k, idx, nullptr, nrm, workspace, knn_ws, stream);
    if (rc != F4L_OK) return rc;
    // The reference's segmentation, label for label, ON THE DEVICE (supervoxel_exact.hip: its sequential fusion and its FIFO
    // exchange as fixed points of parallel passes).  F4L_SV_EXACT_HOST=1, or a cloud whose closures outgrow the device's
    // buffers (F4L_EUNSUPPORTED), replays the sequence on one host core instead: the same labels.
    if (!getenv("F4L_SV_EXACT_HOST")) {
        int32_t nsv_d = 0;
        rc = f4l_supervoxel_segment_exact(xyz, nrm, idx, n, k, resolution, labels_out, &nsv_d, nullptr, workspace, knn_ws, stream);
        if (rc == F4L_OK) {
            if (n_supervoxels_host) *n_supervoxels_host = nsv_d;
            return F4L_OK;
        }
        if (rc != F4L_EUNSUPPORTED) return rc;
        if (getenv("F4L_SV_EXACT_DEBUG")) fprintf(stderr, "[sv exact] device buffers outgrown: replaying on the host\n");
    }
    // the starting lambda's sweep (smallest metric to a neighbour, per point) on the device
    hipLaunchKernelGGL(f4l::sv_min_metric_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xyz, nrm, idx, n, k,
                       resolution, d_dis);
    F4L_LAUNCH_CHECK();
    std::vector<float> h_xyz;
    std::vector<double> h_nrm, h_dis0;
    std::vector<int32_t> h_idx, h_lab;
    try {
        h_xyz.resize((size_t)n * 3);
        h_nrm.resize((size_t)n * 3);
        h_dis0.resize((size_t)n);
        h_idx.resize((size_t)n * k);
        h_lab.resize((size_t)n);
    } catch (const std::bad_alloc &) {
        return F4L_ENOMEM;
    }
    F4L_HIP_CHECK(hipMemcpyAsync(h_xyz.data(), xyz, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipMemcpyAsync(h_nrm.data(), nrm, (size_t)n * 24, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipMemcpyAsync(h_idx.data(), idx, (size_t)n * k * 4, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipMemcpyAsync(h_dis0.data(), d_dis, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    int32_t nsv = 0;
    f4l::BoundaryCtx bctx{xyz, nrm, idx, labels_out, d_flag, d_dis, n, k, resolution, st, F4L_OK};
    f4l::SegmentAssist assist;
    if (!getenv("F4L_SV_HOST_ONLY")) {  // (switch for A/B timing and for the test that both ways give the same labels)
        assist.dis0 = h_dis0.data();
        assist.boundary = f4l::boundary_on_device;
        assist.ctx = &bctx;
    }
    try {
        rc = f4l::segment_host(h_xyz.data(), h_nrm.data(), h_idx.data(), n, k, resolution, h_lab.data(), assist);
    } catch (const std::bad_alloc &) {
        return F4L_ENOMEM;
    }
    if (bctx.rc != F4L_OK) return bctx.rc;
    if (rc < 0) return rc;
    nsv = rc;
    F4L_HIP_CHECK(hipMemcpyAsync(labels_out, h_lab.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    if (n_supervoxels_host) *n_supervoxels_host = nsv;
    return F4L_OK;
}

