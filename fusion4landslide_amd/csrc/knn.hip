// knn.hip -- exact k-nearest neighbours of every point of a cloud (k <= 64), PCA normals, label CSR.
//
// Replaces codelibrary/util/tree/kd_tree.h:266-280 (`KDTree::FindKNearestNeighbors`, called N times in a
// sequential loop at supervoxel.cpp:105-107) and pca_estimate_normals.h:43-108.  The reference walks a
// pointer-based KD-tree one query at a time on one CPU thread; here:
//
//   1. points are binned into a uniform grid: 64-bit linear cell key per point, LSD radix sort of
//      (key, id) pairs (rocPRIM device primitive), run-length encode -> table of occupied cells;
//   2. sorted points are re-laid as float4 {x, y, z, id} so a wave's 64 lanes load 1 KiB contiguous;
//   3. one wavefront per occupied cell: the 3x3 rows of neighbouring cells along x are contiguous runs of
//      the sorted array, located with lane-parallel binary searches over the cell table (one row per
//      lane); every query of the cell then streams those runs, 64 candidates per step, into the
//      wave-resident top-k of topk.h;
//   4. exactness: the k-th distance must not exceed the distance to the faces of the searched block,
//      otherwise the block radius grows by one cell and the query is redone (rare).
// Distances are double with separately rounded mul/add, so d2 and the neighbour order equal the reference's
// except inside groups of exactly equal d2, which are ordered by point id here.
//
// The same kernel answers queries from ANOTHER cloud (f4l_nn_query: the queries are binned into the cloud's grid, outside
// points clamped into its border cells, blocks clipped to the grid and doubled while they hold fewer than k points),
// and the binning machinery doubles as the voxel-grid filter (f4l_voxel_downsample: Open3D's and PCL's cell layouts).
//
// Roofline: algorithmic traffic is 12 B read + 4k B written per point (132 B/pt at k = 30).
// The kernel is bounded by VALU/issue (top-k maintenance), not HBM; bench.py reports the achieved GB/s.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "topk.h"

namespace f4l {

struct GridSpec {
    double minx, miny, minz;
    double inv_h, h;
    int nx, ny, nz;
};

__device__ __forceinline__ void cell_of(const GridSpec &g, float x, float y, float z, int &cx, int &cy, int &cz) {
    cx = (int)(((double)x - g.minx) * g.inv_h);
    cy = (int)(((double)y - g.miny) * g.inv_h);
    cz = (int)(((double)z - g.minz) * g.inv_h);
    cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.ny ? g.ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
}
__device__ __forceinline__ unsigned long long key_of(const GridSpec &g, int cx, int cy, int cz) {
    return ((unsigned long long)cz * (unsigned long long)g.ny + (unsigned long long)cy) * (unsigned long long)g.nx +
           (unsigned long long)cx;
}

// ---- bounding box: per-block partials, finished on the host (tiny) ------------------------------
__global__ void bbox_kernel(const float *__restrict__ xyz, int64_t n, float *__restrict__ partial /* [grid][6] */) {
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = xyz[3 * i + d];
            mn[d] = fminf(mn[d], v);
            mx[d] = fmaxf(mx[d], v);
        }
    }
    __shared__ float sm[4][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], m, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], m, 64));
        }
    }
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { sm[wave][d] = mn[d]; sm[wave][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            partial[6 * blockIdx.x + d] = fminf(fminf(sm[0][d], sm[1][d]), fminf(sm[2][d], sm[3][d]));
            partial[6 * blockIdx.x + 3 + d] = fmaxf(fmaxf(sm[0][3 + d], sm[1][3 + d]), fmaxf(sm[2][3 + d], sm[3][3 + d]));
        }
    }
}

__global__ void cell_key_kernel(const float *__restrict__ xyz, int64_t n, GridSpec g, unsigned long long *__restrict__ keys,
                                int32_t *__restrict__ ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx, cy, cz;
        cell_of(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz);
        keys[i] = key_of(g, cx, cy, cz);
        if (ids) ids[i] = (int32_t)i;
    }
}

__global__ void relayout_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ ids, int64_t n,
                                float4 *__restrict__ sorted) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = ids[i];
        sorted[i] = make_float4(xyz[3 * (int64_t)j], xyz[3 * (int64_t)j + 1], xyz[3 * (int64_t)j + 2], __int_as_float(j));
    }
}

// first index in [0, m) with keys[idx] >= v
__device__ __forceinline__ int lower_bound_u64(const unsigned long long *__restrict__ keys, int m, unsigned long long v) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int KNN_NW = 4;

struct KnnArgs {
    const float4 *sorted;             // [n] {x,y,z,id}
    const unsigned long long *cell_keys;  // [M] ascending
    const int32_t *cell_start;        // [M+1] first sorted index of each occupied cell
    int M;
    // the queries, binned into the cells of the SAME grid (f4l_knn: the cloud itself; f4l_nn_query: another cloud,
    // points outside the grid clamped into its border cells)
    const float4 *q_sorted;
    const unsigned long long *q_cell_keys;
    const int32_t *q_cell_start;
    int Mq;
    int64_t n;
    int k;
    GridSpec g;
    int32_t *idx_out;
    double *d2_out;
};

__global__ __launch_bounds__(KNN_NW * 64) void knn_cells_kernel(KnnArgs a) {
    const int lane = lane_id();
    const int c = (int)blockIdx.x * KNN_NW + (int)(threadIdx.x >> 6);
    if (c >= a.Mq) return;  // whole wave exits together
    const GridSpec g = a.g;
    const unsigned long long key = a.q_cell_keys[c];
    const int cx = (int)(key % (unsigned long long)g.nx);
    const int cy = (int)((key / (unsigned long long)g.nx) % (unsigned long long)g.ny);
    const int cz = (int)(key / ((unsigned long long)g.nx * (unsigned long long)g.ny));
    const int q_begin = a.q_cell_start[c], q_end = a.q_cell_start[c + 1];
    const int k = a.k;
    const int max_dim = max(g.nx, max(g.ny, g.nz));

    // rows of the R = 1 block, one per lane (lanes 0..8), computed once per cell
    int row_lo1 = 0, row_hi1 = 0;
    if (lane < 9) {
        const int yy = cy + (lane % 3) - 1, zz = cz + (lane / 3) - 1;
        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
            const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
            const int ca = lower_bound_u64(a.cell_keys, a.M, key_of(g, x0, yy, zz));
            const int cb = lower_bound_u64(a.cell_keys, a.M, key_of(g, x1, yy, zz) + 1ULL);
            row_lo1 = a.cell_start[ca];
            row_hi1 = a.cell_start[cb];
        }
    }

    for (int q = q_begin; q < q_end; ++q) {
        const float4 qp = a.q_sorted[q];
        const int qid = __float_as_int(qp.w);
        // distance from the query to the faces of its own cell, per axis (conservative by 1e-6 h)
        const double fx = ((double)qp.x - g.minx) - (double)cx * g.h, fy = ((double)qp.y - g.miny) - (double)cy * g.h,
                     fz = ((double)qp.z - g.minz) - (double)cz * g.h;
        WaveTopK best;
        for (int R = 1;;) {
            best.reset();
            bool first = true;  // uniform
            // rows of the block that lie inside the grid (R = 1: the fixed 3 x 3 of row_lo1 / row_hi1)
            const int y0 = cy - R < 0 ? 0 : cy - R, y1 = cy + R >= g.ny ? g.ny - 1 : cy + R;
            const int z0 = cz - R < 0 ? 0 : cz - R, z1 = cz + R >= g.nz ? g.nz - 1 : cz + R;
            const int side = R == 1 ? 3 : y1 - y0 + 1, rows = R == 1 ? 9 : side * (z1 - z0 + 1);
            for (int r0 = 0; r0 < rows; r0 += 64) {
                int lo = 0, hi = 0;
                if (R == 1) { lo = row_lo1; hi = row_hi1; }
                else {
                    const int r = r0 + lane;
                    if (r < rows) {
                        const int yy = y0 + (r % side), zz = z0 + (r / side);
                        {
                            const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R >= g.nx ? g.nx - 1 : cx + R;
                            const int ca = lower_bound_u64(a.cell_keys, a.M, key_of(g, x0, yy, zz));
                            const int cb = lower_bound_u64(a.cell_keys, a.M, key_of(g, x1, yy, zz) + 1ULL);
                            lo = a.cell_start[ca];
                            hi = a.cell_start[cb];
                        }
                    }
                }
                const int nrow = rows - r0 < 64 ? rows - r0 : 64;
                // wider blocks (queries far from the cloud): most rows are empty, visit only the others
                unsigned long long todo = R == 1 ? 0ULL : __ballot(hi > lo);
                for (int rq = 0; R == 1 ? rq < nrow : todo != 0ULL; ++rq) {
                    int rr = rq;
                    if (R != 1) {
                        rr = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ULL;
                    } else {
                        // nearest rows first (own row, the four face neighbours, the four corners): the list tightens
                        // early, and a row that lies beyond the current k-th distance is not streamed at all
                        rr = (int)((0x862075314ULL >> (4 * rq)) & 15ULL);
                        const int dy = rr % 3 - 1, dz = rr / 3 - 1;
                        const double ey = dy < 0 ? fy : (dy > 0 ? g.h - fy : 0.0), ez = dz < 0 ? fz : (dz > 0 ? g.h - fz : 0.0);
                        const double ey0 = ey > 1e-6 * g.h ? ey - 1e-6 * g.h : 0.0, ez0 = ez > 1e-6 * g.h ? ez - 1e-6 * g.h : 0.0;
                        if (ey0 * ey0 + ez0 * ez0 > best.kth(k)) continue;  // uniform (k-th entry is +inf until the list is full)
                    }
                    const int s = __builtin_amdgcn_readlane(lo, rr), e = __builtin_amdgcn_readlane(hi, rr);
                    for (int b = s; b < e; b += 64) {
                        const int ci = b + lane;
                        double cd = __builtin_inf();
                        int cid = 0x7fffffff;
                        if (ci < e) {
                            const float4 cp = a.sorted[ci];
                            cd = dist2_exact(cp.x, cp.y, cp.z, qp.x, qp.y, qp.z);
                            cid = __float_as_int(cp.w);
                        }
                        if (first) { best.fill_sorted(cd, cid); first = false; }
                        else best.offer(cd, cid, k);
                    }
                }
            }
            // exactness: k-th distance strictly inside the searched block (faces at the grid border do not count)
            double margin = __builtin_inf();
            const double eps = 1e-6 * g.h;
            if (cx - R > 0) margin = fmin(margin, fx + (double)R * g.h - eps);
            if (cx + R < g.nx - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fx - eps);
            if (cy - R > 0) margin = fmin(margin, fy + (double)R * g.h - eps);
            if (cy + R < g.ny - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fy - eps);
            if (cz - R > 0) margin = fmin(margin, fz + (double)R * g.h - eps);
            if (cz + R < g.nz - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fz - eps);
            const double dk = best.kth(k);
            if (dk < margin * margin || R >= max_dim) break;  // uniform: dk, margin are wave-uniform
            // fewer than k points in the whole block (queries far from the cloud): double it instead of one more layer
            R = dk == __builtin_inf() ? (2 * R < max_dim ? 2 * R : max_dim) : R + 1;
        }
        if (lane < k) {
            a.idx_out[(int64_t)qid * k + lane] = best.i;
            if (a.d2_out) a.d2_out[(int64_t)qid * k + lane] = best.d;
        }
    }
}

// ---- PCA normals (pca_estimate_normals.h:43-108, unit weights), one thread per point ----------------
#pragma clang fp contract(off)
__global__ void normals_kernel(const float *__restrict__ xyz, int64_t n, const int32_t *__restrict__ knn, int k,
                               double *__restrict__ normals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *nb = knn + i * k;
    double cx = 0.0, cy = 0.0, cz = 0.0, sum = 0.0;
    for (int j = 0; j < k; ++j) {
        const int64_t q = nb[j];
        cx += (double)xyz[3 * q]; cy += (double)xyz[3 * q + 1]; cz += (double)xyz[3 * q + 2];
        sum += 1.0;
    }
    const double inv = 1.0 / sum;
    cx *= inv; cy *= inv; cz *= inv;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, ws = 0;
    for (int j = 0; j < k; ++j) {
        const int64_t q = nb[j];
        const double x = (double)xyz[3 * q] - cx, y = (double)xyz[3 * q + 1] - cy, z = (double)xyz[3 * q + 2] - cz;
        a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
        ws += 1.0;
    }
    const double t = 1.0 / ws;
    a00 *= t; a01 *= t; a02 *= t; a11 *= t; a12 *= t; a22 *= t;
    const double q = (a00 + a11 + a22) / 3.0;
    double pq = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) +
                2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
    pq = sqrt(pq / 6.0);
    const double mpq = pow(1.0 / pq, 3.0);
    const double det_b = mpq * ((a00 - q) * ((a11 - q) * (a22 - q) - a12 * a12) - a01 * (a01 * (a22 - q) - a12 * a02) +
                                a02 * (a01 * a12 - (a11 - q) * a02));
    const double r = 0.5 * det_b;
    double phi;
    if (r <= -1.0) phi = 3.14159265358979323846 / 3.0;
    else if (r >= 1.0) phi = 0.0;
    else phi = acos(r) / 3.0;
    const double eig = q + 2.0 * pq * cos(phi + 3.14159265358979323846 * (2.0 / 3.0));
    double nx = a01 * a12 - a02 * (a11 - eig);
    double ny = a01 * a02 - a12 * (a00 - eig);
    double nz = (a00 - eig) * (a11 - eig) - a01 * a01;
    const double norm = sqrt(nx * nx + ny * ny + nz * nz);
    if (norm == 0.0) { nx = 0.0; ny = 0.0; nz = 1.0; }
    else { const double s = 1.0 / norm; nx *= s; ny *= s; nz *= s; }
    normals[3 * i] = nx; normals[3 * i + 1] = ny; normals[3 * i + 2] = nz;
}

__global__ void iota_kernel(int32_t *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}
__global__ void label_hist_kernel(const int32_t *__restrict__ labels, int64_t n, int64_t K, unsigned long long *__restrict__ hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t l = labels[i];
        if (l >= 0 && l < K) atomicAdd(&hist[l + 1], 1ULL);
    }
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline unsigned grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t b = (n + block - 1) / block;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// workspace carve for f4l_knn
struct KnnWs {
    unsigned long long *keys_a, *keys_b, *cell_keys;
    int32_t *ids_a, *ids_b, *cell_counts, *cell_start, *n_cells;
    float4 *sorted;
    float *bbox_partial;
    void *prim_temp;
    size_t prim_bytes, total;
};

static int knn_ws_layout(int64_t n, KnnWs &w, unsigned char *base) {
    size_t sort_b = 0, rle_b = 0, scan_b = 0;
    unsigned long long *k0 = nullptr;
    int32_t *i0 = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, sort_b, k0, k0, i0, i0, (size_t)n, 0, 64, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::run_length_encode(nullptr, rle_b, k0, (unsigned int)n, k0, i0, i0, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::exclusive_scan(nullptr, scan_b, i0, i0, 0, (size_t)n + 1, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    size_t prim = sort_b > rle_b ? sort_b : rle_b;
    prim = prim > scan_b ? prim : scan_b;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    w.keys_a = (unsigned long long *)carve((size_t)n * 8);
    w.keys_b = (unsigned long long *)carve((size_t)n * 8);
    w.cell_keys = (unsigned long long *)carve((size_t)n * 8);
    w.ids_a = (int32_t *)carve((size_t)n * 4);
    w.ids_b = (int32_t *)carve((size_t)n * 4);
    w.cell_counts = (int32_t *)carve(((size_t)n + 1) * 4);
    w.cell_start = (int32_t *)carve(((size_t)n + 1) * 4);
    w.n_cells = (int32_t *)carve(256);
    w.sorted = (float4 *)carve((size_t)n * 16);
    w.bbox_partial = (float *)carve(256 * 6 * 4);
    w.prim_temp = carve(prim);
    w.prim_bytes = prim;
    w.total = o;
    return F4L_OK;
}

}  // namespace f4l

extern "C" size_t f4l_knn_workspace_bytes(int64_t n, int k) {
    (void)k;
    if (n <= 0) return 0;
    f4l::KnnWs w;
    if (f4l::knn_ws_layout(n, w, nullptr) != F4L_OK) return 0;
    return w.total;
}

namespace f4l {
// Steps 1-4 of f4l_knn: bounding box, cell size for ~k/2 points per occupied cell, points sorted by cell, occupied-cell
// table.  Synchronises `st`.
static int bbox_to_host(const float *xyz, int64_t n, float *partial, hipStream_t st, double *mn, double *mx) {
    const unsigned bb_grid = grid_for(n, 256, 256);
    hipLaunchKernelGGL(bbox_kernel, dim3(bb_grid), dim3(256), 0, st, xyz, n, partial);
    F4L_LAUNCH_CHECK();
    float hb[256 * 6];
    F4L_HIP_CHECK(hipMemcpyAsync(hb, partial, (size_t)bb_grid * 6 * 4, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    for (int d = 0; d < 3; ++d) { mn[d] = 1e300; mx[d] = -1e300; }
    for (unsigned b = 0; b < bb_grid; ++b)
        for (int d = 0; d < 3; ++d) {
            if (hb[6 * b + d] < mn[d]) mn[d] = hb[6 * b + d];
            if (hb[6 * b + 3 + d] > mx[d]) mx[d] = hb[6 * b + 3 + d];
        }
    for (int d = 0; d < 3; ++d)
        if (!(mx[d] >= mn[d]) || !std::isfinite(mn[d]) || !std::isfinite(mx[d])) return F4L_EINVAL;  // NaN / inf coordinates
    return F4L_OK;
}

static int knn_build_grid(const float *xyz, int64_t n, int k, KnnWs &w, hipStream_t st, GridSpec &g, int &M) {
    // 1. bounding box
    double mn[3], mx[3];
    {
        const int rc = bbox_to_host(xyz, n, w.bbox_partial, st, mn, mx);
        if (rc != F4L_OK) return rc;
    }

    // 2. cell size: aim at ~k/2 points per occupied cell; start from a surface-density guess and correct with
    //    the measured occupancy (the result is exact for any h, only speed depends on it)
    double ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    double e[3] = {ext[0], ext[1], ext[2]};
    std::sort(e, e + 3);
    const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    const double target = k / 2.0 < 4.0 ? 4.0 : k / 2.0;
    double h;
    if (diag == 0.0) h = 1.0;
    else {
        const double area = (e[2] > 0 ? e[2] : diag) * (e[1] > 0 ? e[1] : (e[2] > 0 ? e[2] : diag) * 1e-3);
        h = std::sqrt(target * area / (double)n);
        if (!(h > 0.0)) h = diag;
    }
    M = 0;
    for (int iter = 0; iter < 5; ++iter) {
        // keep every axis below 2^20 cells so the linear key fits comfortably in 63 bits
        const double hmin = (e[2] > 0 ? e[2] : 1.0) / 1048000.0;
        if (h < hmin) h = hmin;
        g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
        g.h = h; g.inv_h = 1.0 / h;
        g.nx = (int)(ext[0] / h) + 1; g.ny = (int)(ext[1] / h) + 1; g.nz = (int)(ext[2] / h) + 1;
        hipLaunchKernelGGL(cell_key_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, g, w.keys_a, w.ids_a);
        F4L_LAUNCH_CHECK();
        const double ncell = (double)g.nx * (double)g.ny * (double)g.nz;
        int end_bit = 1;
        while (end_bit < 63 && (double)(1ULL << end_bit) < ncell) ++end_bit;
        size_t tb = w.prim_bytes;
        F4L_HIP_CHECK(rocprim::radix_sort_pairs(w.prim_temp, tb, w.keys_a, w.keys_b, w.ids_a, w.ids_b, (size_t)n, 0,
                                                (unsigned)end_bit, st, false));
        tb = w.prim_bytes;
        F4L_HIP_CHECK(rocprim::run_length_encode(w.prim_temp, tb, w.keys_b, (unsigned int)n, w.cell_keys, w.cell_counts,
                                                 w.n_cells, st, false));
        F4L_HIP_CHECK(hipMemcpyAsync(&M, w.n_cells, 4, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipStreamSynchronize(st));
        if (M <= 0) return F4L_EHIP;
        const double occ = (double)n / (double)M;
        if (diag == 0.0 || (occ >= 0.6 * target && occ <= 1.7 * target) || iter == 4) break;
        // occupancy scales like h^dim with dim between 2 (surface) and 3 (volume); 2.5 converges for both
        double f = std::pow(target / occ, 1.0 / 2.5);
        f = f < 0.25 ? 0.25 : (f > 4.0 ? 4.0 : f);
        h *= f;
    }
    // 3. cell_start = exclusive scan of the run lengths (M+1 entries)
    F4L_HIP_CHECK(hipMemsetAsync(w.cell_counts + M, 0, 4, st));
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim_temp, tb, w.cell_counts, w.cell_start, 0, (size_t)M + 1,
                                          rocprim::plus<int32_t>(), st, false));
    // 4. sorted float4 layout
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, w.ids_b, n, w.sorted);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
}  // namespace f4l

// Synchronises `stream` (the bounding box and the occupied-cell count are read back to size the grid).
extern "C" int f4l_knn(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, void *workspace,
                       size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (!xyz || n <= 0 || k < 1 || k > n || !idx_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    KnnWs w;
    int rc = knn_ws_layout(n, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    GridSpec g;
    int M = 0;
    rc = knn_build_grid(xyz, n, k, w, st, g, M);
    if (rc != F4L_OK) return rc;
    // 5. one wave per occupied cell
    KnnArgs a;
    a.sorted = w.sorted; a.cell_keys = w.cell_keys; a.cell_start = w.cell_start; a.M = M; a.n = n; a.k = k; a.g = g;
    a.q_sorted = w.sorted; a.q_cell_keys = w.cell_keys; a.q_cell_start = w.cell_start; a.Mq = M;
    a.idx_out = idx_out; a.d2_out = d2_out;
    hipLaunchKernelGGL(knn_cells_kernel, dim3((unsigned)((M + KNN_NW - 1) / KNN_NW)), dim3(KNN_NW * 64), 0, st, a);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- k nearest points of ANOTHER cloud --------------------------------------------------------------------------
// (the cKDTree queries around the hot loop: src/coarse_to_fine_matching_base.py:1042-1046 `_voxel_subsampling`,
//  :2716-2754 `_compute_median_resolution` is the self-query f4l_knn)
extern "C" size_t f4l_nn_query_workspace_bytes(int64_t n, int64_t m, int k) {
    (void)k;
    if (n <= 0 || m <= 0) return 0;
    f4l::KnnWs w, wq;
    if (f4l::knn_ws_layout(n, w, nullptr) != F4L_OK || f4l::knn_ws_layout(m, wq, nullptr) != F4L_OK) return 0;
    return w.total + wq.total;
}

// Synchronises `stream` (grid sizing reads the bounding box and the cell counts back).
extern "C" int f4l_nn_query(const float *cloud, int64_t n, const float *queries, int64_t m, int k, int32_t *idx_out,
                            double *d2_out, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (m == 0) return F4L_OK;
    if (!cloud || !queries || n <= 0 || m < 0 || k < 1 || k > n || !idx_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n > 0x7fffffffLL || m > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    KnnWs w, wq;
    int rc = knn_ws_layout(n, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    rc = knn_ws_layout(m, wq, (unsigned char *)workspace + w.total);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total + wq.total) return F4L_EWORKSPACE;
    GridSpec g;
    int M = 0, Mq = 0;
    rc = knn_build_grid(cloud, n, k, w, st, g, M);
    if (rc != F4L_OK) return rc;
    // the queries, binned into the cells of the cloud's grid (outside points land in its border cells; the search
    // measures true distances, the cell only says where to start)
    {
        double mn[3], mx[3];
        rc = bbox_to_host(queries, m, wq.bbox_partial, st, mn, mx);  // (rejects NaN / inf queries)
        if (rc != F4L_OK) return rc;
    }
    hipLaunchKernelGGL(cell_key_kernel, dim3(grid_for(m)), dim3(256), 0, st, queries, m, g, wq.keys_a, wq.ids_a);
    F4L_LAUNCH_CHECK();
    const double ncell = (double)g.nx * (double)g.ny * (double)g.nz;
    int end_bit = 1;
    while (end_bit < 63 && (double)(1ULL << end_bit) < ncell) ++end_bit;
    size_t tb = wq.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_pairs(wq.prim_temp, tb, wq.keys_a, wq.keys_b, wq.ids_a, wq.ids_b, (size_t)m, 0,
                                            (unsigned)end_bit, st, false));
    tb = wq.prim_bytes;
    F4L_HIP_CHECK(rocprim::run_length_encode(wq.prim_temp, tb, wq.keys_b, (unsigned int)m, wq.cell_keys, wq.cell_counts,
                                             wq.n_cells, st, false));
    F4L_HIP_CHECK(hipMemcpyAsync(&Mq, wq.n_cells, 4, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    if (Mq <= 0) return F4L_EHIP;
    F4L_HIP_CHECK(hipMemsetAsync(wq.cell_counts + Mq, 0, 4, st));
    tb = wq.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(wq.prim_temp, tb, wq.cell_counts, wq.cell_start, 0, (size_t)Mq + 1,
                                          rocprim::plus<int32_t>(), st, false));
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(m)), dim3(256), 0, st, queries, wq.ids_b, m, wq.sorted);
    F4L_LAUNCH_CHECK();
    KnnArgs a;
    a.sorted = w.sorted; a.cell_keys = w.cell_keys; a.cell_start = w.cell_start; a.M = M; a.n = n; a.k = k; a.g = g;
    a.q_sorted = wq.sorted; a.q_cell_keys = wq.cell_keys; a.q_cell_start = wq.cell_start; a.Mq = Mq;
    a.idx_out = idx_out; a.d2_out = d2_out;
    hipLaunchKernelGGL(knn_cells_kernel, dim3((unsigned)((Mq + KNN_NW - 1) / KNN_NW)), dim3(KNN_NW * 64), 0, st, a);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- voxel grid filter ---------------------------------------------------------------------------------------------
// Open3D PointCloud::VoxelDownSample as called at src/coarse_to_fine_matching_base.py:1024-1025 [3P-knowledge]:
// voxel index = floor((p - (min_bound - voxel/2)) / voxel) in double, one output point per occupied voxel = the mean of
// its points.  Open3D emits voxels in hash-map order (unpinned); here: ascending (z, y, x) voxel index, and the mean
// sums a voxel's points in ascending input index (deterministic).
namespace f4l {
__global__ void voxel_key_kernel(const float *__restrict__ xyz, int64_t n, double minx, double miny, double minz,
                                 double voxel, unsigned long long nx, unsigned long long ny,
                                 unsigned long long *__restrict__ keys, int32_t *__restrict__ ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long cx = (unsigned long long)floor(((double)xyz[3 * i] - minx) / voxel);
        const unsigned long long cy = (unsigned long long)floor(((double)xyz[3 * i + 1] - miny) / voxel);
        const unsigned long long cz = (unsigned long long)floor(((double)xyz[3 * i + 2] - minz) / voxel);
        keys[i] = (cz * ny + cy) * nx + cx;
        ids[i] = (int32_t)i;
    }
}
// pcl::VoxelGrid's cell of a point: floor(p * inverse_leaf) - min_b, all in float32, min_b = floor(min_p * inverse_leaf)
__global__ void voxel_key_pcl_kernel(const float *__restrict__ xyz, int64_t n, float inv_leaf, int bx, int by, int bz,
                                     unsigned long long nx, unsigned long long ny, unsigned long long *__restrict__ keys,
                                     int32_t *__restrict__ ids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long cx = (unsigned long long)((int)floorf(xyz[3 * i] * inv_leaf) - bx);
        const unsigned long long cy = (unsigned long long)((int)floorf(xyz[3 * i + 1] * inv_leaf) - by);
        const unsigned long long cz = (unsigned long long)((int)floorf(xyz[3 * i + 2] * inv_leaf) - bz);
        keys[i] = (cz * ny + cy) * nx + cx;
        ids[i] = (int32_t)i;
    }
}
#pragma clang fp contract(off)
__global__ void voxel_mean_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ ids,
                                  const int32_t *__restrict__ start, int M, double *__restrict__ pts_out,
                                  int32_t *__restrict__ count_out, int32_t *__restrict__ voxel_of_point) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= M) return;
    const int s = start[v], e = start[v + 1];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int j = s; j < e; ++j) {  // the radix sort is stable: ascending input index inside a voxel
        const int64_t i = ids[j];
        sx += (double)xyz[3 * i]; sy += (double)xyz[3 * i + 1]; sz += (double)xyz[3 * i + 2];
        if (voxel_of_point) voxel_of_point[i] = v;
    }
    const double c = (double)(e - s);
    pts_out[3 * (int64_t)v] = sx / c; pts_out[3 * (int64_t)v + 1] = sy / c; pts_out[3 * (int64_t)v + 2] = sz / c;
    if (count_out) count_out[v] = e - s;
}
}  // namespace f4l

extern "C" size_t f4l_voxel_downsample_workspace_bytes(int64_t n) { return f4l_knn_workspace_bytes(n, 1); }

// pts_out: room for n points (3 doubles each); *m_out (host) receives the number of voxels.  Synchronises `stream`.
extern "C" int f4l_voxel_downsample(const float *xyz, int64_t n, double voxel, int layout, double *pts_out,
                                    int32_t *count_out, int32_t *voxel_of_point_out, int64_t *m_out, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (!m_out || (layout != F4L_VOXEL_OPEN3D && layout != F4L_VOXEL_PCL)) return F4L_EINVAL;
    *m_out = 0;
    if (n == 0) return F4L_OK;
    if (!xyz || n < 0 || !(voxel > 0.0) || !pts_out || !workspace) return F4L_EINVAL;
    if (n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    KnnWs w;
    int rc = knn_ws_layout(n, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    double mn[3], mx[3];
    rc = bbox_to_host(xyz, n, w.bbox_partial, st, mn, mx);
    if (rc != F4L_OK) return rc;
    double dims[3];
    if (layout == F4L_VOXEL_OPEN3D) {
        for (int d = 0; d < 3; ++d) {
            mn[d] -= 0.5 * voxel;
            dims[d] = std::floor((mx[d] - mn[d]) / voxel) + 1.0;
            if (dims[d] > 2097151.0) return F4L_EUNSUPPORTED;  // 3 x 21 bits of key (Open3D: "voxel_size is too small")
        }
        hipLaunchKernelGGL(voxel_key_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, mn[0], mn[1], mn[2], voxel,
                           (unsigned long long)dims[0], (unsigned long long)dims[1], w.keys_a, w.ids_a);
    } else {
        // pcl::VoxelGrid::applyFilter [3P-knowledge]: float32 leaf and inverse leaf, cells counted from floor(min * inv)
        const float inv_leaf = 1.0f / (float)voxel;
        int b[3];
        for (int d = 0; d < 3; ++d) {
            b[d] = (int)std::floor((float)mn[d] * inv_leaf);
            dims[d] = (double)((int)std::floor((float)mx[d] * inv_leaf) - b[d] + 1);
            if (dims[d] > 2097151.0) return F4L_EUNSUPPORTED;  // (PCL: "Leaf size is too small for the input dataset")
        }
        hipLaunchKernelGGL(voxel_key_pcl_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, inv_leaf, b[0], b[1], b[2],
                           (unsigned long long)dims[0], (unsigned long long)dims[1], w.keys_a, w.ids_a);
    }
    F4L_LAUNCH_CHECK();
    const double ncell = dims[0] * dims[1] * dims[2];
    int end_bit = 1;
    while (end_bit < 64 && std::ldexp(1.0, end_bit) < ncell) ++end_bit;
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_pairs(w.prim_temp, tb, w.keys_a, w.keys_b, w.ids_a, w.ids_b, (size_t)n, 0,
                                            (unsigned)end_bit, st, false));
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::run_length_encode(w.prim_temp, tb, w.keys_b, (unsigned int)n, w.cell_keys, w.cell_counts,
                                             w.n_cells, st, false));
    int M = 0;
    F4L_HIP_CHECK(hipMemcpyAsync(&M, w.n_cells, 4, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    if (M <= 0) return F4L_EHIP;
    F4L_HIP_CHECK(hipMemsetAsync(w.cell_counts + M, 0, 4, st));
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim_temp, tb, w.cell_counts, w.cell_start, 0, (size_t)M + 1,
                                          rocprim::plus<int32_t>(), st, false));
    hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, xyz, w.ids_b, w.cell_start,
                       M, pts_out, count_out, voxel_of_point_out);
    F4L_LAUNCH_CHECK();
    *m_out = M;
    return F4L_OK;
}

extern "C" int f4l_normals(const float *xyz, int64_t n, const int32_t *knn_idx, int k, double *normals_out, void *stream) {
    if (!xyz || n <= 0 || !knn_idx || k < 1 || !normals_out) return F4L_EINVAL;
    hipLaunchKernelGGL(f4l::normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n,
                       knn_idx, k, normals_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- labels -> CSR ---------------------------------------------------------------------------------
namespace f4l {
struct CsrWs {
    int32_t *keys_out, *iota;
    unsigned long long *hist;
    void *prim_temp;
    size_t prim_bytes, total;
};
static int csr_ws_layout(int64_t n, int64_t K, CsrWs &w, unsigned char *base) {
    size_t sort_b = 0, scan_b = 0;
    int32_t *i0 = nullptr;
    unsigned long long *u0 = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, sort_b, i0, i0, i0, i0, (size_t)n, 0, 32, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::inclusive_scan(nullptr, scan_b, u0, u0, (size_t)K + 1, rocprim::plus<unsigned long long>(), 0, false) != hipSuccess)
        return F4L_EHIP;
    const size_t prim = sort_b > scan_b ? sort_b : scan_b;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    w.keys_out = (int32_t *)carve((size_t)n * 4);
    w.iota = (int32_t *)carve((size_t)n * 4);
    w.hist = (unsigned long long *)carve(((size_t)K + 1) * 8);
    w.prim_temp = carve(prim);
    w.prim_bytes = prim;
    w.total = o;
    return F4L_OK;
}
}  // namespace f4l

extern "C" size_t f4l_labels_to_csr_workspace_bytes(int64_t n, int64_t K) {
    if (n <= 0 || K <= 0) return 0;
    f4l::CsrWs w;
    if (f4l::csr_ws_layout(n, K, w, nullptr) != F4L_OK) return 0;
    return w.total;
}

extern "C" int f4l_labels_to_csr(const int32_t *labels, int64_t n, int64_t K, int32_t *order_out, int64_t *off_out,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (n < 0 || K <= 0 || !off_out || (n > 0 && (!labels || !order_out || !workspace))) return F4L_EINVAL;
    if (n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        F4L_HIP_CHECK(hipMemsetAsync(off_out, 0, ((size_t)K + 1) * 8, st));
        return F4L_OK;
    }
    CsrWs w;
    int rc = csr_ws_layout(n, K, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(256), 0, st, w.iota, n);
    F4L_LAUNCH_CHECK();
    int end_bit = 1;
    while (end_bit < 31 && (1LL << end_bit) < K) ++end_bit;
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_pairs(w.prim_temp, tb, labels, w.keys_out, w.iota, order_out, (size_t)n, 0,
                                            (unsigned)end_bit, st, false));  // LSD radix sort is stable
    F4L_HIP_CHECK(hipMemsetAsync(w.hist, 0, ((size_t)K + 1) * 8, st));
    hipLaunchKernelGGL(label_hist_kernel, dim3(grid_for(n)), dim3(256), 0, st, labels, n, K, w.hist);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::inclusive_scan(w.prim_temp, tb, w.hist, reinterpret_cast<unsigned long long *>(off_out),
                                          (size_t)K + 1, rocprim::plus<unsigned long long>(), st, false));
    return F4L_OK;
}
