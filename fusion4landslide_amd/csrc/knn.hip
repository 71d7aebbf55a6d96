// knn.hip -- exact k-nearest neighbours of every point of a cloud (k <= 64), PCA normals, label CSR.
//
// Replaces codelibrary/util/tree/kd_tree.h:266-280 (`KDTree::FindKNearestNeighbors`, called N times in a
// sequential loop at supervoxel.cpp:105-107) and pca_estimate_normals.h:43-108.  The reference walks a
// pointer-based KD-tree one query at a time on one CPU thread; here:
//
//   1. points are binned into a uniform grid: linear cell key per point (32-bit when the grid allows), radix sort of
//      (key, id) pairs (rocPRIM onesweep passes), run-length encode -> table of occupied cells, plus ONE WORD PER CELL of
//      the whole grid while that stays within two words per point (`dense`: a row of cells is two loads, else two binary
//      searches over the occupied cells);
//   2. sorted points are re-laid as float4 {x, y, z, id} so a wave's 64 lanes load 1 KiB contiguous;
//   3. the search proper, one of three kernels:
//        knn_lanes_kernel   k <= 36: one LANE per query; a wave takes 64 consecutive sorted points and streams their common
//                           candidate set (the 3 x 3 rows of cells around them) through an LDS tile; per-lane d2 histogram,
//                           survivors sorted on registers with exact d2, PCA normal fused (f4l_knn_normals);
//        nn_small_kernel    k <= 4: one lane per query walking the rows of cells of its OWN block from the sorted array
//                           (the 1-NN / 2-NN of the label transfer and of the median resolution);
//        knn_cells_kernel / knn_listed_kernel   one WAVE per query with the wave-resident top-k of topk.h: k > 36, and
//                           the queries the lane kernel could not certify;
//   4. exactness: the k-th distance must lie inside the searched block of cells, otherwise the block grows (straight to
//      the radius the k-th distance found so far asks for) and the query is redone.
// Distances are double with separately rounded mul/add, so d2 and the neighbour order equal the reference's
// except inside groups of exactly equal d2, which are ordered by point id here.
//
// The same kernels answer queries from ANOTHER cloud (f4l_nn_query: the queries are binned into the cloud's grid, outside
// points clamped into its border cells, blocks clipped to the grid and doubled while they hold fewer than k points),
// and the binning machinery doubles as the voxel-grid filter (f4l_voxel_downsample: Open3D's and PCL's cell layouts).
//
// Roofline: algorithmic traffic is 12 B read + 4k B written per point (132 B/pt at k = 30).
// The kernels are bounded by VALU issue, not HBM; bench.py reports the achieved GB/s.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "select.h"
#include "topk.h"

namespace f4l {

// rocprim's radix sort switches to a merge sort at or below 2^20 items (147 us for the 1 M cell keys of a tile: ten merge
// passes); with the cell keys' few significant bits the onesweep passes are three and take a third of that.
using KeySortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;

struct GridSpec {
    double minx, miny, minz;
    double inv_h, h;
    double inv_hz;  // = inv_h, or 0 for a flat cloud (one layer of cells along z: nz = 1, every point in layer 0)
    int nx, ny, nz;
};

// The grid as the DEVICE sized it (f4l_knn's no-host-round-trip path): kernels that are handed one take the grid and the number
// of occupied cells from it instead of from their launch arguments.
struct DevGrid {
    GridSpec g;
    int M;         // occupied cells
    int dense_ok;  // the grid is small enough for the one-word-per-cell table
};

__device__ __forceinline__ void cell_of(const GridSpec &g, float x, float y, float z, int &cx, int &cy, int &cz) {
    cx = (int)(((double)x - g.minx) * g.inv_h);
    cy = (int)(((double)y - g.miny) * g.inv_h);
    cz = (int)(((double)z - g.minz) * g.inv_hz);
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

// the same keys as 32-bit words (grids of fewer than 2^32 cells: nearly all): half the bytes through the radix sort
__global__ void cell_key32_kernel(const float *__restrict__ xyz, int64_t n, GridSpec g, unsigned int *__restrict__ keys,
                                  int32_t *__restrict__ ids, const DevGrid *__restrict__ dg = nullptr) {
    if (dg) g = dg->g;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx, cy, cz;
        cell_of(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz);
        keys[i] = (unsigned int)key_of(g, cx, cy, cz);
        ids[i] = (int32_t)i;
    }
}
__global__ void widen_keys_kernel(const unsigned int *__restrict__ k32, const int32_t *__restrict__ count, unsigned long long *__restrict__ k64) {
    const int m = *count;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < m; i += (int)(gridDim.x * blockDim.x)) k64[i] = k32[i];
}

__global__ void relayout_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ ids, int64_t n,
                                float4 *__restrict__ sorted) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = ids[i];
        const float x = xyz[3 * (int64_t)j], y = xyz[3 * (int64_t)j + 1], z = xyz[3 * (int64_t)j + 2];
        sorted[i] = make_float4(x, y, z, __int_as_float(j));
    }
}

// (cell_runs_kernel, below) Per occupied cell c = (cx, cy, cz) and r = (dy + 1) + 3 (dz + 1): the run of the sorted array that holds the cells
// cx - 1 .. cx + 1 of row (cy + dy, cz + dz): run_lo[9 c + r] = first point with key >= key(cx - 1, ..), run_hi[9 c + r] = first
// point with key > key(cx + 1, ..) (both 0 for a row outside the grid).  A wave whose queries span the cells c_first .. c_last
// of one row takes [run_lo[c_first], run_hi[c_last]).  Also the cell index of every sorted point.

// first index in [0, m) with keys[idx] >= v
__device__ __forceinline__ int lower_bound_u64(const unsigned long long *__restrict__ keys, int m, unsigned long long v) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// [lo, hi) of the sorted array for the cells with keys k0 .. k1 (one row of cells along x).  `dense` (when the grid is small
// enough for one word per cell, see knn_build_grid): dense[c] = number of points in cells with keys below c -- two loads
// instead of two binary searches over the occupied cells.
__device__ __forceinline__ void cell_range(const int32_t *__restrict__ dense, const unsigned long long *__restrict__ cell_keys,
                                           const int32_t *__restrict__ cell_start, int M, unsigned long long k0, unsigned long long k1,
                                           int &lo, int &hi) {
    if (dense) { lo = dense[k0]; hi = dense[k1 + 1ULL]; }
    else { lo = cell_start[lower_bound_u64(cell_keys, M, k0)]; hi = cell_start[lower_bound_u64(cell_keys, M, k1 + 1ULL)]; }
}
__global__ void dense_scatter_kernel(const unsigned long long *__restrict__ cell_keys, const int32_t *__restrict__ cell_counts, int M,
                                     int32_t *__restrict__ dense, const DevGrid *__restrict__ dg = nullptr) {
    if (dg) {
        if (!dg->dense_ok) return;
        M = dg->M;
    }
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j < M) dense[cell_keys[j] + 1ULL] = cell_counts[j];
}

constexpr int KNN_NW = 4;

// (q_keys / q_start / Mq: the occupied cells of the QUERIES -- the cloud's own table for f4l_knn, the query cloud's, binned into
//  the same grid, for f4l_nn_query; cell_keys / cell_start / M: the cloud searched)
__global__ void cell_runs_kernel(const unsigned long long *__restrict__ q_keys, const int32_t *__restrict__ q_start, int Mq,
                                 const unsigned long long *__restrict__ cell_keys, const int32_t *__restrict__ cell_start, int M,
                                 const int32_t *__restrict__ dense, GridSpec g, int32_t *__restrict__ run_lo, int32_t *__restrict__ run_hi,
                                 int32_t *__restrict__ pcell, const DevGrid *__restrict__ dg = nullptr) {
    if (dg) {
        g = dg->g;
        M = Mq = dg->M;
        if (!dg->dense_ok) dense = nullptr;
    }
    const int64_t t64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t64 >= 9 * (int64_t)Mq) return;
    const int t = (int)t64;
    const int c = t / 9, r = t % 9;
    const unsigned long long key = q_keys[c];
    const int cx = (int)(key % (unsigned long long)g.nx);
    const int cy = (int)((key / (unsigned long long)g.nx) % (unsigned long long)g.ny);
    const int cz = (int)(key / ((unsigned long long)g.nx * (unsigned long long)g.ny));
    const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
    int lo = 0, hi = 0;
    if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
        const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
        cell_range(dense, cell_keys, cell_start, M, key_of(g, x0, yy, zz), key_of(g, x1, yy, zz), lo, hi);
    }
    run_lo[t] = lo;
    run_hi[t] = hi;
    if (r == 4)  // (dy, dz) = (0, 0): this thread also tags the cell's own queries
        for (int i = q_start[c]; i < q_start[c + 1]; ++i) pcell[i] = c;
}

struct KnnArgs {
    const float4 *sorted;             // [n] {x,y,z,id}
    const unsigned long long *cell_keys;  // [M] ascending
    const int32_t *cell_start;        // [M+1] first sorted index of each occupied cell
    int M;
    const int32_t *dense;             // [cells + 1] points below each cell of the grid, or nullptr (cell_range)
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
    double *nn1_out = nullptr;        // [n] squared distance to the nearest OTHER point (slot 1 of the row), or null
    const DevGrid *dg;                // non-null: grid, cell counts and the dense-table switch come from the device (see DevGrid)
    // POSITION mode (the partition's own neighbour search, f4l_partition_neighbours): a point is named by its position in the
    // cell-sorted array, not by the caller's index -- neighbours, ties between equal distances, the rows of nn1_out and of the
    // normals -- and idx_out is TRANSPOSED, idx_out[j * n + position]: 64 consecutive queries of a wave write 256 contiguous
    // bytes per neighbour slot, and the segmentation works in this order as it stands (no relabelling pass, no transpose)
    int pos_mode = 0;
};
__device__ __forceinline__ void adopt_device_grid(KnnArgs &a) {
    if (!a.dg) return;
    a.g = a.dg->g;
    a.M = a.Mq = a.dg->M;
    if (!a.dg->dense_ok) a.dense = nullptr;
}

// Rows of the R = 1 block of cell (cx, cy, cz), one per lane (lanes 0..8): [lo, hi) of the sorted array.
__device__ __forceinline__ void block1_rows(const KnnArgs &a, int cx, int cy, int cz, int &row_lo1, int &row_hi1) {
    const GridSpec &g = a.g;
    const int lane = lane_id();
    row_lo1 = 0; row_hi1 = 0;
    if (lane < 9) {
        const int yy = cy + (lane % 3) - 1, zz = cz + (lane / 3) - 1;
        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
            const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
            cell_range(a.dense, a.cell_keys, a.cell_start, a.M, key_of(g, x0, yy, zz), key_of(g, x1, yy, zz), row_lo1, row_hi1);
        }
    }
}

// One query answered by one wavefront: streams the block's rows nearest first into the wave-resident top-k, grows the
// block until the k-th distance lies inside it.  (cx, cy, cz) is the query's cell, row_lo1 / row_hi1 from block1_rows.
__device__ __forceinline__ void knn_query_wave(const KnnArgs &a, const float4 qp, int cx, int cy, int cz, int row_lo1,
                                               int row_hi1, WaveTopK &best) {
    const GridSpec &g = a.g;
    const int lane = lane_id();
    const int k = a.k;
    const int max_dim = max(g.nx, max(g.ny, g.nz));
    // distance from the query to the faces of its own cell, per axis (conservative by 1e-6 h)
    const double fx = ((double)qp.x - g.minx) - (double)cx * g.h, fy = ((double)qp.y - g.miny) - (double)cy * g.h,
                 fz = ((double)qp.z - g.minz) - (double)cz * g.h;
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
                    const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R >= g.nx ? g.nx - 1 : cx + R;
                    cell_range(a.dense, a.cell_keys, a.cell_start, a.M, key_of(g, x0, yy, zz), key_of(g, x1, yy, zz), lo, hi);
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
                        cid = a.pos_mode ? ci : __float_as_int(cp.w);
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
        // fewer than k points in the whole block (queries far from the cloud): double it instead of one more layer; else
        // straight to the first block whose faces lie beyond the k-th distance found (it can only shrink: that block ends the
        // search -- queries displaced against the cloud, e.g. a moving slope's epoch 2 against epoch 1)
        if (dk == __builtin_inf()) R = 2 * R < max_dim ? 2 * R : max_dim;
        else {
            const double need = (sqrt(dk) + eps) * g.inv_h + 1.0;
            const int Rj = need < (double)max_dim ? (int)need : max_dim;
            R = Rj > R + 1 ? Rj : R + 1;
        }
    }
}

// One wavefront per occupied query cell, its queries one after the other (f4l_nn_query, and f4l_knn for k > KR_MAX_K).
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
    int row_lo1, row_hi1;
    block1_rows(a, cx, cy, cz, row_lo1, row_hi1);
    for (int q = q_begin; q < q_end; ++q) {
        const float4 qp = a.q_sorted[q];
        const int qid = __float_as_int(qp.w);
        WaveTopK best;
        knn_query_wave(a, qp, cx, cy, cz, row_lo1, row_hi1, best);
        if (lane < k) {
            a.idx_out[(int64_t)qid * k + lane] = best.i;
            if (a.d2_out) a.d2_out[(int64_t)qid * k + lane] = best.d;
            if (a.nn1_out && lane == 1) a.nn1_out[qid] = best.d;
        }
    }
}

// The queries the one-lane-per-query kernel below could not certify (listed by their position in the sorted array):
// one wavefront per query, the search above.
__global__ __launch_bounds__(KNN_NW * 64) void knn_listed_kernel(KnnArgs a, const int32_t *__restrict__ list,
                                                                const int32_t *__restrict__ count) {
    adopt_device_grid(a);
    const int lane = lane_id();
    const int n_list = *count;
    const int k = a.k;
    const GridSpec g = a.g;
    for (int i = (int)blockIdx.x * KNN_NW + (int)(threadIdx.x >> 6); i < n_list; i += (int)gridDim.x * KNN_NW) {
        const float4 qp = a.q_sorted[list[i]];
        const int qid = __float_as_int(qp.w);
        int cx, cy, cz;
        cell_of(g, qp.x, qp.y, qp.z, cx, cy, cz);
        int row_lo1, row_hi1;
        block1_rows(a, cx, cy, cz, row_lo1, row_hi1);
        WaveTopK best;
        knn_query_wave(a, qp, cx, cy, cz, row_lo1, row_hi1, best);
        if (lane < k) {
            if (a.pos_mode) {
                a.idx_out[(int64_t)lane * a.n + list[i]] = best.i;
                if (a.nn1_out && lane == 1) a.nn1_out[list[i]] = best.d;
            } else {
                a.idx_out[(int64_t)qid * k + lane] = best.i;
                if (a.d2_out) a.d2_out[(int64_t)qid * k + lane] = best.d;
                if (a.nn1_out && lane == 1) a.nn1_out[qid] = best.d;
            }
        }
    }
}

// ---- a handful of neighbours (k <= KS_MAX_K): one LANE per query, candidates straight from the sorted array ------------------
// The 2-NN of `_compute_median_resolution`, the 1-NN of the label transfer and of `_voxel_subsampling`.  With cells of ~4
// points a query's own 3 x 3 block holds ~36 candidates, but a wave of the lane kernel below shares ONE candidate set, the
// union of its 64 queries' blocks (~220 points at this cell size): 0.40 ms per 1 M queries whatever k.  Here every lane walks
// the rows of cells of its own block (queries come in sorted order: neighbouring lanes read overlapping ranges of the sorted
// array, from L1), measures exactly (dist2_exact) and keeps its k nearest by (d2, index) in registers; the block grows by
// the rule of knn_query_wave until the k-th distance lies inside it -- a query displaced against the cloud (a moving slope's
// second epoch: 0.2-0.5 m against cells of 0.125 m) ends with a block of ~11 x 11 cells = ~500 candidates.  1 M queries, k = 1,
// a quarter of them displaced: 0.25 ms (lane kernel + one wave per uncertified query: 0.40 + 2.9 ms in round 2b).
// `list` = nullptr: all nq queries of q_sorted; else the listed ones.
constexpr int KS_MAX_K = 4, KS_BATCH = 4;
// Round 5: the walk is two-stage.  Every candidate used to cost an exact double distance and an insertion by (d2, index) into four
// slots (~150 SIMD cycles a candidate and wave: 2.2 ms per 10 M queries whatever k).  Now the block is scanned in float32 -- the
// differences of neighbouring float coordinates are exact, three roundings remain: d2 to 2e-7 of itself -- keeping the KK + 1
// smallest approximate d2 with the candidates' positions; when the (k + 1)-th approximate d2 lies beyond the k-th by more than
// that error can explain (a factor 1 + 2e-6, and 1e-30 for the flush-to-zero range), the k nearest ARE the first k slots as a set,
// and only they are measured exactly and ordered by (d2, index).  A lane whose slots k and k + 1 are closer than that -- duplicates,
// lattices, genuine near-ties -- walks its block again the old way; results are those of the exact walk in every case.
// KK = slots kept (1, 2 or 4 >= k).
template <int KK>
__device__ __forceinline__ void ks_insert_exact(double (&bd)[KK], int (&bi)[KK], double d, int id) {
#pragma unroll
    for (int t = 0; t < KK; ++t) {  // insertion, ascending by (d2, index); the last falls off
        const bool lt = d < bd[t] || (d == bd[t] && id < bi[t]);
        const double dn = lt ? bd[t] : d;
        const int in = lt ? bi[t] : id;
        bd[t] = lt ? d : bd[t];
        bi[t] = lt ? id : bi[t];
        d = dn; id = in;
    }
}
template <int KK>
__global__ __launch_bounds__(256) void nn_small_kernel(KnnArgs a, int nq, const int32_t *__restrict__ list, const int32_t *__restrict__ count) {
    adopt_device_grid(a);
    const int n_q = list ? *count : nq;
    const GridSpec g = a.g;
    const int k = a.k;
    const int max_dim = max(g.nx, max(g.ny, g.nz));
    const double eps = 1e-6 * g.h;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n_q; i += (int)(gridDim.x * blockDim.x)) {
        const float4 qp = a.q_sorted[list ? list[i] : i];
        const int qid = __float_as_int(qp.w);
        int cx, cy, cz;
        cell_of(g, qp.x, qp.y, qp.z, cx, cy, cz);
        const double fx = ((double)qp.x - g.minx) - (double)cx * g.h, fy = ((double)qp.y - g.miny) - (double)cy * g.h,
                     fz = ((double)qp.z - g.minz) - (double)cz * g.h;
        double bd[KK];
        int bi[KK];
        // From the second round on the previous round's k-th distance bounds the answer: only the cells a ball of that radius around
        // the query reaches are walked -- rows of cells farther than it in y / z are skipped, the others clipped in x (a query displaced
        // against the cloud by a few cells, a moving slope's second epoch, used to walk the whole square block again: 9 x 9 cells
        // where the disc covers a fifth of them).  What lies in the block outside the ball is farther than the k candidates already
        // known, so the walk's result is that of the whole block, and the block's certificate below stands.
        double lim = __builtin_inf();
        const double qxr = fx + (double)cx * g.h;  // the query's x in the grid's frame
        auto gap = [&](double f, int c, int cc) {  // distance along one axis from the query (offset f in its cell c) to cell cc
            const double d = cc > c ? (double)(cc - c) * g.h - f : (cc < c ? f - (double)(cc + 1 - c) * g.h : 0.0);
            return d > 0.0 ? d : 0.0;
        };
        for (int R = 1;;) {
            const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R >= g.nx ? g.nx - 1 : cx + R;
            const int y0 = cy - R < 0 ? 0 : cy - R, y1 = cy + R >= g.ny ? g.ny - 1 : cy + R;
            const int z0 = cz - R < 0 ? 0 : cz - R, z1 = cz + R >= g.nz ? g.nz - 1 : cz + R;
            auto walk = [&](auto &&visit) {
                for (int zz = z0; zz <= z1; ++zz) {
                    const double gz = g.nz > 1 ? gap(fz, cz, zz) : 0.0;
                    if (gz * gz > lim) continue;
                    for (int yy = y0; yy <= y1; ++yy) {
                        const double gy = gap(fy, cy, yy);
                        const double rem = lim - gz * gz - gy * gy;
                        if (rem < 0.0) continue;
                        int xa = x0, xb = x1;
                        if (lim < __builtin_inf()) {
                            const double w = sqrt(rem) + eps;
                            const double lo_c = floor((qxr - w) * g.inv_h), hi_c = floor((qxr + w) * g.inv_h);
                            xa = lo_c > (double)x0 ? (int)lo_c : x0;
                            xb = hi_c < (double)x1 ? (int)hi_c : x1;
                            if (xa > xb) continue;
                        }
                        int lo, hi;
                        cell_range(a.dense, a.cell_keys, a.cell_start, a.M, key_of(g, xa, yy, zz), key_of(g, xb, yy, zz), lo, hi);
                        for (int j = lo; j < hi; j += KS_BATCH) {  // (KS_BATCH loads in flight per lane instead of one dependent chain)
                            float4 c[KS_BATCH];
#pragma unroll
                            for (int u = 0; u < KS_BATCH; ++u) c[u] = a.sorted[j + u < hi ? j + u : hi - 1];
#pragma unroll
                            for (int u = 0; u < KS_BATCH; ++u) visit(c[u], j + u, j + u < hi);
                        }
                    }
                }
            };
            // stage 1: the KK + 1 smallest approximate d2 of the block, with their positions in the sorted array
            float fd[KK + 1];
            int fj[KK + 1];
#pragma unroll
            for (int t = 0; t <= KK; ++t) { fd[t] = __builtin_inff(); fj[t] = -1; }
            walk([&](const float4 &cp, int j, bool real) {
                const float dx = cp.x - qp.x, dy = cp.y - qp.y, dz = cp.z - qp.z;
                float d = real ? __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) : __builtin_inff();
                int jj = j;
#pragma unroll
                for (int t = 0; t <= KK; ++t) {
                    const bool lt = d < fd[t];
                    const float dn = lt ? fd[t] : d;
                    const int jn = lt ? fj[t] : jj;
                    fd[t] = lt ? d : fd[t];
                    fj[t] = lt ? jj : fj[t];
                    d = dn; jj = jn;
                }
            });
            // slots k - 1 and k (0-based): is the gap between them more than the float32 error of either?
            float f_k1 = fd[0], f_k = fd[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                if (t == k - 1) f_k1 = fd[t];
                if (t == k) f_k = fd[t];
            }
#pragma unroll
            for (int j = 0; j < KK; ++j) { bd[j] = __builtin_inf(); bi[j] = 0x7fffffff; }
            const bool separated = f_k1 == __builtin_inff() || !(f_k <= f_k1 * 1.000002f + 1e-30f);  // (fewer than k candidates: nothing to order)
            if (separated) {
                // stage 2: the first k slots measured exactly, ordered by (d2, index)
#pragma unroll
                for (int t = 0; t < KK; ++t)
                    if (t < k && fj[t] >= 0) {
                        const float4 cp = a.sorted[fj[t]];
                        ks_insert_exact<KK>(bd, bi, dist2_exact(cp.x, cp.y, cp.z, qp.x, qp.y, qp.z), __float_as_int(cp.w));
                    }
            } else {
                // near-ties around the k-th neighbour: the exact walk
                walk([&](const float4 &cp, int, bool real) {
                    ks_insert_exact<KK>(bd, bi, real ? dist2_exact(cp.x, cp.y, cp.z, qp.x, qp.y, qp.z) : __builtin_inf(),
                                        real ? __float_as_int(cp.w) : 0x7fffffff);
                });
            }
            double dk = bd[0];
#pragma unroll
            for (int t = 1; t < KK; ++t)
                if (t == k - 1) dk = bd[t];
            // exactness: the k-th distance lies strictly inside the searched block (faces at the grid border do not count)
            double margin = __builtin_inf();
            if (cx - R > 0) margin = fmin(margin, fx + (double)R * g.h - eps);
            if (cx + R < g.nx - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fx - eps);
            if (cy - R > 0) margin = fmin(margin, fy + (double)R * g.h - eps);
            if (cy + R < g.ny - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fy - eps);
            if (cz - R > 0) margin = fmin(margin, fz + (double)R * g.h - eps);
            if (cz + R < g.nz - 1) margin = fmin(margin, ((double)(R + 1)) * g.h - fz - eps);
            if (dk < margin * margin || R >= max_dim) break;
            if (dk == __builtin_inf()) R = 2 * R < max_dim ? 2 * R : max_dim;
            else {
                const double need = (sqrt(dk) + eps) * g.inv_h + 1.0;
                const int Rj = need < (double)max_dim ? (int)need : max_dim;
                R = Rj > R + 1 ? Rj : R + 1;
                lim = dk * (1.0 + 1e-12);
            }
        }
        if (a.nn1_out) {  // (f4l_epoch_join: only the distance to the nearest other point is wanted; k = 2 there)
            if constexpr (KK >= 2) a.nn1_out[qid] = bd[1];
        }
        if (a.idx_out) {
#pragma unroll
            for (int t = 0; t < KK; ++t)
                if (t < k) {
                    a.idx_out[(int64_t)qid * k + t] = bi[t];
                    if (a.d2_out) a.d2_out[(int64_t)qid * k + t] = bd[t];
                }
        }
    }
}
static void launch_nn_small(const KnnArgs &a, int64_t m, hipStream_t st) {
    const dim3 grid((unsigned)((m + 255) / 256)), block(256);
    const int32_t *none = nullptr;
    if (a.k == 1) hipLaunchKernelGGL(nn_small_kernel<1>, grid, block, 0, st, a, (int)m, none, none);
    else if (a.k == 2) hipLaunchKernelGGL(nn_small_kernel<2>, grid, block, 0, st, a, (int)m, none, none);
    else hipLaunchKernelGGL(nn_small_kernel<KS_MAX_K>, grid, block, 0, st, a, (int)m, none, none);
}

// ---- fast path of f4l_knn: one LANE per query -------------------------------------------------------------------------
// A wavefront takes 64 CONSECUTIVE points of the sorted array.  They lie in a few x-adjacent cells of one (y, z) row of
// cells (a wave that straddles the end of a row takes one turn per row), so they share one candidate set: the 3 x 3 rows
// around theirs, from one cell before the first query's cell to one cell after the last one's -- nine contiguous runs of
// the sorted array.  The wave streams those runs through a 1 KB tile of its own in LDS (64 candidates per coalesced load,
// made relative to the wave's first query in double and rounded to float32 once, by the loading lane) and every lane
// measures each candidate against its own query with one broadcast ds_read_b128 and float32 arithmetic:
//   pass 1  histogram of the approximate d2 per lane, in LDS (32 bins, a quarter octave of d2 each: scale free, a bin near
//           the k-th neighbour holds ~ 0.19 k points whatever the local density);
//           -> the first bin T at which the count reaches k;
//   pass 2  the candidates below bin T's upper edge, widened by far more than the float32 error of a wave-relative
//           coordinate (1e-4 relative + 1e-6 h^2), go to the lane's list in LDS: a superset of the k nearest, k plus a handful;
//   sort    the list is loaded into registers with the EXACT d2 (double, from the float coordinates, separately rounded
//           multiply and add: what the reference computes) and sorted by a bitonic network on registers -- no cross-lane
//           traffic, no divergence; exact ties by point index.
// A query is certified when its k-th distance lies inside the searched block (same test as the wave-per-query search);
// the few that are not (sparse borders, more survivors than the list holds, fewer than k candidates) are listed and redone
// by knn_listed_kernel.
// (Candidates through scalar loads -- coordinates in SGPRs, no LDS -- were measured first: 2.4 ms per 1 M points; the scalar
// cache misses to L2 one line at a time and nothing overlaps them.)
constexpr int KR_NB = 32;      // histogram bins
constexpr int KR_CAP = 43;     // survivors a lane can keep
constexpr int KR_MAX_K = 36;   // largest k the fast path takes (room for the threshold bin's overshoot)
constexpr int KR_NW = 4;       // waves per workgroup (8 KB histogram + 11.25 KB list + 0.8 KB tile each: two workgroups per CU)

struct KnnLanesArgs {
    KnnArgs a;
    const int32_t *run_lo, *run_hi;  // [9 M] candidate runs per occupied cell (cell_runs_kernel)
    const int32_t *pcell;            // [n] cell index of every sorted point
    int32_t *fb_list;   // sorted positions of the queries to redo
    int32_t *fb_count;
    const float *xyz;   // the cloud in input order (fused normals)
    double *normals_out;
    int bin_base;       // bin of d2 = (its float image >> 21) - bin_base, clamped to [0, KR_NB)
    float edge_slack;   // 1e-6 h^2
    unsigned long long *prof;  // profiling build (-DF4L_KNN_PROF): cycles per phase, summed over the waves
};
#ifdef F4L_KNN_PROF
#define KR_TICK(slot)                                                                              \
    do {                                                                                           \
        const unsigned long long now__ = __builtin_readcyclecounter();                             \
        if (lane_id() == 0 && ra.prof) atomicAdd(&ra.prof[slot], now__ - tick__);                  \
        tick__ = __builtin_readcyclecounter();                                                     \
    } while (0)
#else
#define KR_TICK(slot) do { } while (0)
#endif

__device__ __forceinline__ double dist2_d(double ax, double ay, double az, double bx, double by, double bz) {
#pragma clang fp contract(off)
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    double t = dx * dx;
    t = t + dy * dy;
    t = t + dz * dz;
    return t;
}
// compare-exchange on registers, ascending; equal keys keep their places (their order is settled afterwards)
__device__ __forceinline__ void kr_ce(double &ka, int &pa, double &kb, int &pb) {
    const bool sw = kb < ka;
    const double lo = __builtin_fmin(ka, kb), hi = __builtin_fmax(ka, kb);  // (no NaNs here: distances and +inf padding)
    const int plo = sw ? pb : pa, phi = sw ? pa : pb;
    ka = lo; kb = hi; pa = plo; pb = phi;
}
// pca_estimate_normals.h:43-108 on k points held in registers (the arithmetic of normals_kernel)
template <int MAXK>
__device__ __forceinline__ void pca_normal_regs(const float (&px)[MAXK], const float (&py)[MAXK], const float (&pz)[MAXK], int k,
                                                double &nx, double &ny, double &nz) {
#pragma clang fp contract(off)
    double cx = 0.0, cy = 0.0, cz = 0.0, sum = 0.0;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        if (j < k) { cx += (double)px[j]; cy += (double)py[j]; cz += (double)pz[j]; sum += 1.0; }
    const double inv = 1.0 / sum;
    cx *= inv; cy *= inv; cz *= inv;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, ws = 0;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        if (j < k) {
            const double x = (double)px[j] - cx, y = (double)py[j] - cy, z = (double)pz[j] - cz;
            a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
            ws += 1.0;
        }
    const double t = 1.0 / ws;
    a00 *= t; a01 *= t; a02 *= t; a11 *= t; a12 *= t; a22 *= t;
    const double q = (a00 + a11 + a22) / 3.0;
    double pq = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
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
    nx = a01 * a12 - a02 * (a11 - eig);
    ny = a01 * a02 - a12 * (a00 - eig);
    nz = (a00 - eig) * (a11 - eig) - a01 * a01;
    const double norm = sqrt(nx * nx + ny * ny + nz * nz);
    if (norm == 0.0) { nx = 0.0; ny = 0.0; nz = 1.0; }
    else { const double s = 1.0 / norm; nx *= s; ny *= s; nz *= s; }
}

// Walks the nine candidate runs of a block (lane r < 9 holds run r as [lo, hi)) through the wave's LDS tile, KR_U candidates
// per step, coordinates relative to the wave origin.  The tile is three arrays (x, y, z) so that one broadcast ds_read_b128
// brings the same coordinate of four candidates and packed float32 instructions work on them in pairs.  Every step is split
// in two so that LDS latency stays off the critical path at two waves per SIMD:
//     R = measure(index of the first, x[KR_U], y[KR_U], z[KR_U], how many are real)      registers only (places past the
//                                                                                        run hold points at infinity)
//     <the NEXT step's tile reads are issued here>
//     commit(R)                                                                           the step's LDS writes / atomics
// (LDS operations of a wave complete in order: reads issued before the commit's writes are not held up by them.)
// The next 64 candidates are requested from global memory before the current 64 are used.
constexpr int KR_U = 8;
// Lanes of a wave that share a candidate set: the wave's 64 queries are KR_G groups of KR_L consecutive points, every group
// with the nine runs of ITS OWN block -- the cells its queries span, plus one on either side -- streamed through its own
// segment of the tile.  The union block of all 64 queries holds 324 candidates per lane at 1 M points; the larger of two
// half-wave blocks 242, of four quarter-wave blocks 201 (simulated on the C2 cloud: tools/knn_block_sim.py): a wave walks
// as many steps as its longest group.
#ifndef F4L_KNN_GROUPS
#define F4L_KNN_GROUPS 4
#endif
// KR_T: candidates a group stages per tile; a lane loads KR_T / KR_L of them.  One per lane is what works: two loads in flight
// per lane (tiles of 32 for four groups, 64 for two) cost 50 % -- the prefetched registers are copied, i.e. waited for,
// right behind the loads -- and so measured: per 10 M points 4.96 ms one group (round 2), 4.54 with padded tiles, 4.22 two
// groups, 4.17 four groups of 16; 6.2 - 6.6 ms with the longer tiles.
#ifndef F4L_KNN_TILE
#define F4L_KNN_TILE 16
#endif
constexpr int KR_G = F4L_KNN_GROUPS, KR_L = 64 / KR_G, KR_T = KR_L < F4L_KNN_TILE ? F4L_KNN_TILE : KR_L, KR_PL = KR_T / KR_L;
constexpr int KR_TS = 3 * (KR_T + 4);  // tile segment of a group: x[], y[], z[] of KR_T + 4
struct KrStep { float4 x[KR_U / 4], y[KR_U / 4], z[KR_U / 4]; };
__device__ __forceinline__ void kr_read(const float *gt, int u, KrStep &t) {
#pragma unroll
    for (int v = 0; v < KR_U / 4; ++v) {
        t.x[v] = *reinterpret_cast<const float4 *>(gt + u + 4 * v);
        t.y[v] = *reinterpret_cast<const float4 *>(gt + (KR_T + 4) + u + 4 * v);
        t.z[v] = *reinterpret_cast<const float4 *>(gt + 2 * (KR_T + 4) + u + 4 * v);
    }
}
// lo / hi: lane (first lane of its group) + r holds run r of the group's block as [lo, hi); groups without a block hold
// empty runs.  Everything that is uniform inside a group lives in vector registers (the groups differ); the wave iterates
// until its last group is through, groups that are done measure points at infinity.
template <class R, class F1, class F2>
__device__ __forceinline__ void kr_walk(const float4 *__restrict__ sorted, float *tile /* [KR_G][KR_TS] */, int lo, int hi, double ox,
                                        double oy, double oz, F1 &&measure, F2 &&commit) {
    const int lane = lane_id(), l = lane & (KR_L - 1), gb = lane & ~(KR_L - 1);
    float *gt = tile + (lane / KR_L) * KR_TS;
    // tiles of <= KR_T candidates per group, run after run; the load of the NEXT tile (of this run or of the next non-empty
    // one) is in flight while the current tile is consumed: nothing waits for global memory except the very first tile
    unsigned int runs = (unsigned int)(__ballot(lo < hi) >> gb) & 0x1ffu;  // the group's non-empty runs
    int r = runs ? __builtin_ctz(runs) : 9;
    int base = __shfl(lo, gb + (r < 9 ? r : 0), 64), e = __shfl(hi, gb + (r < 9 ? r : 0), 64);
    if (!__any(r < 9)) return;
    float4 nxt[KR_PL];
    bool nxt_real[KR_PL];
#pragma unroll
    for (int c = 0; c < KR_PL; ++c) {
        const int j = base + l + c * KR_L;
        nxt[c] = sorted[r < 9 ? (j < e ? j : e - 1) : 0];
        nxt_real[c] = r < 9 && j < e;
    }
    while (__any(r < 9)) {
        // places of the tile beyond the run hold a point at infinity (d2 = +inf: last bin of pass 1, below no edge in pass 2),
        // so that neither pass has to ask which of a step's candidates are real
#pragma unroll
        for (int c = 0; c < KR_PL; ++c) {
            gt[l + c * KR_L] = nxt_real[c] ? (float)((double)nxt[c].x - ox) : 1e30f;
            gt[KR_T + 4 + l + c * KR_L] = nxt_real[c] ? (float)((double)nxt[c].y - oy) : 1e30f;
            gt[2 * (KR_T + 4) + l + c * KR_L] = nxt_real[c] ? (float)((double)nxt[c].z - oz) : 1e30f;
        }
        const int cbase = base, m = r < 9 ? (e - base < KR_T ? e - base : KR_T) : 0;
        base += KR_T;
        const bool next_run = r < 9 && base >= e;
        if (next_run) {
            runs &= ~((2u << r) - 1u);
            r = runs ? __builtin_ctz(runs) : 9;
        }
        {
            const int s2 = __shfl(lo, gb + (r < 9 ? r : 0), 64), e2 = __shfl(hi, gb + (r < 9 ? r : 0), 64);
            base = next_run ? s2 : base;
            e = next_run ? e2 : e;
        }
        // (unconditional on purpose: a load under `if (r < 9)` makes the compiler copy the loaded registers into the loop's own
        //  right behind the load, i.e. wait for it at once, and the tile's latency is back on the critical path)
#pragma unroll
        for (int c = 0; c < KR_PL; ++c) {
            const int j = base + l + c * KR_L;
            nxt[c] = sorted[r < 9 ? (j < e ? j : e - 1) : 0];
            nxt_real[c] = r < 9 && j < e;
        }
        // two register sets used in turn (a single set would be copied from the prefetched one every step)
        KrStep t0, t1;
        kr_read(gt, 0, t0);
        int mw = 0;  // the longest group's count: the wave's trip count, in a scalar register
#pragma unroll
        for (int g = 0; g < KR_G; ++g) { const int mg = __builtin_amdgcn_readlane(m, g * KR_L); mw = mg > mw ? mg : mw; }
        for (int u = 0; u < mw; u += 2 * KR_U) {
            {
                const R res = measure(cbase + u, t0, m - u);
                __builtin_amdgcn_sched_barrier(0);
                if (KR_U < KR_T && u + KR_U < mw) kr_read(gt, u + KR_U, t1);
                __builtin_amdgcn_sched_barrier(0);
                commit(res);
            }
            if (KR_U < KR_T && u + KR_U < mw) {
                const R res = measure(cbase + u + KR_U, t1, m - u - KR_U);
                __builtin_amdgcn_sched_barrier(0);
                if (2 * KR_U < KR_T && u + 2 * KR_U < mw) kr_read(gt, u + 2 * KR_U, t0);
                __builtin_amdgcn_sched_barrier(0);
                commit(res);
            }
        }
    }
}

template <bool POS>
__global__ __launch_bounds__(KR_NW * 64, 2) void knn_lanes_kernel(KnnLanesArgs ra) {
    __shared__ unsigned int s_hist[KR_NW][KR_NB * 64];
    __shared__ unsigned int s_list[KR_NW][(KR_CAP + 1) * 64];  // (+ 1: the row predicated-off writes of pass 2 land in)
    __shared__ __attribute__((aligned(16))) float s_tile[KR_NW][KR_G * KR_TS];
#ifdef F4L_KNN_LDS_PAD  // (measurement only: LDS asked for and never used -- one workgroup per CU instead of two: what occupancy is worth)
    __shared__ unsigned int s_pad[F4L_KNN_LDS_PAD / 4];
    if (ra.a.n < 0) s_pad[threadIdx.x] = 1u;
    if (ra.a.n < -1) s_list[0][0] = s_pad[threadIdx.x ^ 1];
#endif
    KnnArgs a = ra.a;
    int bin_base = ra.bin_base;
    float edge_slack = ra.edge_slack;
    if (a.dg) {  // (the host did not know the cell size when it launched)
        adopt_device_grid(a);
        edge_slack = (float)(1e-6 * a.g.h * a.g.h);
        bin_base = (int)(__float_as_uint((float)(4.0 * a.g.h * a.g.h)) >> 21) - (KR_NB - 1);
    }
    const GridSpec g = a.g;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    unsigned int *hist = s_hist[wave] + lane, *list = s_list[wave] + lane;
    float *tile = s_tile[wave];
    const int64_t q = ((int64_t)blockIdx.x * KR_NW + wave) * 64 + lane;
    const bool valid = q < a.n;
    const int k = a.k;
#ifdef F4L_KNN_PROF
    unsigned long long tick__ = __builtin_readcyclecounter();
#endif
    const float4 qp = a.q_sorted[valid ? q : 0];
    const int pc = ra.pcell[valid ? q : 0];
    int cx, cy, cz;
    cell_of(g, qp.x, qp.y, qp.z, cx, cy, cz);
    const double qx = (double)qp.x, qy = (double)qp.y, qz = (double)qp.z;
    // wave origin: the first query (the wave's queries and candidates lie within a few cells of it)
    const double ox = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.x))),
                 oy = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.y))),
                 oz = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qp.z)));
    const float rx = (float)(qx - ox), ry_ = (float)(qy - oy), rz_ = (float)(qz - oz);
#pragma unroll
    for (int b = 0; b < KR_NB; ++b) hist[b * 64] = 0u;
    int cnt = 0;
    bool fb = false;
    unsigned long long remaining = __ballot(valid);
    const int gl = lane & (KR_L - 1), gb = lane & ~(KR_L - 1);
    const unsigned long long gmask = KR_L == 64 ? ~0ULL : ((1ULL << (KR_L & 63)) - 1ULL);
    while (remaining != 0ULL) {  // one turn per (y, z) row of cells among a group's queries: nearly always one
        // every group takes the row of its first query that is still waiting (a group that is through sits the turn out)
        const unsigned long long grem = (remaining >> gb) & gmask;
        const int leader = gb + (grem ? __ffsll((long long)grem) - 1 : 0);
        const int ry = __shfl(cy, leader, 64), rz = __shfl(cz, leader, 64);
        const bool mine = valid && grem != 0ULL && cy == ry && cz == rz;
        const unsigned long long act = __ballot(mine);
        remaining &= ~act;
        // the sorted order is x-fastest: the first / last lane of the row sit in its first / last cell; the runs come from
        // the per-cell table (cell_runs_kernel)
        const unsigned long long gact = (act >> gb) & gmask;
        const int c_first = __shfl(pc, gb + (gact ? __ffsll((long long)gact) - 1 : 0), 64);
        const int c_last = __shfl(pc, gb + (gact ? 63 - __clzll((long long)gact) : 0), 64);
        int lo = 0, hi = 0;
        if (gl < 9 && gact != 0ULL) {
            lo = ra.run_lo[9 * c_first + gl];
            hi = ra.run_hi[9 * c_last + gl];
        }
        const unsigned int one = mine ? 1u : 0u;
        KR_TICK(0);
        // pass 1: per-lane histogram of the approximate d2 over the block's candidates
        struct Bins { int addr[KR_U]; };
        kr_walk<Bins>(a.sorted, tile, lo, hi, ox, oy, oz,
            [&](int, const KrStep &t, int) {
                Bins o;
#pragma unroll
                for (int v = 0; v < KR_U / 4; ++v) {
                    const float xs[4] = {t.x[v].x, t.x[v].y, t.x[v].z, t.x[v].w}, ys[4] = {t.y[v].x, t.y[v].y, t.y[v].z, t.y[v].w},
                                zs[4] = {t.z[v].x, t.z[v].y, t.z[v].z, t.z[v].w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float dx = xs[w] - rx, dy = ys[w] - ry_, dz = zs[w] - rz_;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        int b = (int)(__float_as_uint(d2) >> 21) - bin_base;
                        b = b < 0 ? 0 : (b > KR_NB - 1 ? KR_NB - 1 : b);
                        o.addr[4 * v + w] = b * 64;
                    }
                }
                return o;
            },
            [&](const Bins &o) {
#pragma unroll
                for (int w = 0; w < KR_U; ++w) atomicAdd(&hist[o.addr[w]], one);
            });
        KR_TICK(1);
        // the first bin at which the count reaches k
        int T = KR_NB;
        unsigned int cum = 0u;
#pragma unroll
        for (int b = 0; b < KR_NB; ++b) {
            const unsigned int h = hist[b * 64];
            hist[b * 64] = 0u;
            if (T == KR_NB && cum + h >= (unsigned int)k) T = b;
            cum += h;
        }
        if (mine && T == KR_NB) fb = true;  // fewer than k points in the block
        // upper edge of bin T, widened far beyond the float32 error of the wave-relative arithmetic; lanes of other rows: nothing
        float edge = T >= KR_NB - 1 ? __builtin_inff() : __uint_as_float((unsigned int)(T + bin_base + 1) << 21) * 1.0001f + edge_slack;
        edge = mine ? edge : -1.0f;
        KR_TICK(2);
        // pass 2: the candidates below it go to the lane's list (branch free: a lane that does not take a candidate writes
        // it to the spare row)
        struct Takes { int row[KR_U]; unsigned int index; };
        kr_walk<Takes>(a.sorted, tile, lo, hi, ox, oy, oz,
            [&](int c_index, const KrStep &t, int) {
                Takes o;
                o.index = (unsigned int)c_index;
#pragma unroll
                for (int v = 0; v < KR_U / 4; ++v) {
                    const float xs[4] = {t.x[v].x, t.x[v].y, t.x[v].z, t.x[v].w}, ys[4] = {t.y[v].x, t.y[v].y, t.y[v].z, t.y[v].w},
                                zs[4] = {t.z[v].x, t.z[v].y, t.z[v].z, t.z[v].w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float dx = xs[w] - rx, dy = ys[w] - ry_, dz = zs[w] - rz_;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const bool take = d2 < edge;
                        const int at = cnt < KR_CAP ? cnt : KR_CAP;
                        o.row[4 * v + w] = (take ? at : KR_CAP) * 64;
                        cnt += take ? 1 : 0;
                    }
                }
                return o;
            },
            [&](const Takes &o) {
#pragma unroll
                for (int w = 0; w < KR_U; ++w) list[o.row[w]] = o.index + (unsigned int)w;
            });
    }
    KR_TICK(3);
    if (cnt > KR_CAP) { fb = true; cnt = KR_CAP; }

    // the survivors, exact d2 from the float coordinates (dist2_exact: what the reference computes), sorted on registers
    double key[KR_CAP];
    int pay[KR_CAP];
#pragma unroll
    for (int j = 0; j < KR_CAP; ++j) {
        const bool ok = j < cnt;
        const unsigned int slot = ok ? list[j * 64] : 0u;
        const float4 p = a.sorted[slot];
        const double d = dist2_exact(p.x, p.y, p.z, qp.x, qp.y, qp.z);
        key[j] = ok ? d : __builtin_inf();
        pay[j] = ok ? (POS ? (int)slot : __float_as_int(p.w)) : 0x7fffffff;
    }
    KR_TICK(4);
    // bitonic network of 64 with ascending comparators only (per merge size one mirrored stage, then the half-cleaners).
    // Entries KR_CAP..63 would be +inf padding; in an all-ascending network the largest elements at the top never move, so
    // every comparator that touches them is a no-op and is left out (and the padding needs no registers).
#pragma unroll
    for (int lm = 1; lm <= 6; ++lm) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int l = i ^ ((1 << lm) - 1);
            if (l > i && l < KR_CAP) kr_ce(key[i], pay[i], key[l < KR_CAP ? l : 0], pay[l < KR_CAP ? l : 0]);
        }
#pragma unroll
        for (int lj = lm - 2; lj >= 0; --lj) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int l = i ^ (1 << lj);
                if (l > i && l < KR_CAP) kr_ce(key[i], pay[i], key[l < KR_CAP ? l : 0], pay[l < KR_CAP ? l : 0]);
            }
        }
    }
    KR_TICK(5);
    // exactly equal distances (gridded or duplicated points): order by point index, like the wave-resident list
    bool anyeq = false;
#pragma unroll
    for (int j = 0; j + 1 < KR_CAP; ++j) anyeq = anyeq | ((key[j] == key[j + 1]) & (pay[j] > pay[j + 1]));  // (padding: equal indices)
    if (__ballot(anyeq) != 0ULL) {  // rare: odd-even transposition inside groups of equal keys until nothing moves
        for (int pass = 0; pass < KR_CAP; ++pass) {
            bool moved = false;
#pragma unroll
            for (int j = 0; j + 1 < KR_CAP; j += 2)
            {
                const bool sw = (key[j] == key[j + 1]) & (pay[j] > pay[j + 1]);
                const int lo = sw ? pay[j + 1] : pay[j], hi = sw ? pay[j] : pay[j + 1];
                pay[j] = lo; pay[j + 1] = hi; moved = moved | sw;
            }
#pragma unroll
            for (int j = 1; j + 1 < KR_CAP; j += 2) {
                const bool sw = (key[j] == key[j + 1]) & (pay[j] > pay[j + 1]);
                const int lo = sw ? pay[j + 1] : pay[j], hi = sw ? pay[j] : pay[j + 1];
                pay[j] = lo; pay[j + 1] = hi; moved = moved | sw;
            }
            if (__ballot(moved) == 0ULL) break;
        }
    }
    // certified when the k-th distance lies strictly inside the searched block (faces at the grid border do not count)
    double dk = __builtin_inf();
#pragma unroll
    for (int j = 0; j < KR_MAX_K; ++j)
        if (j == k - 1) dk = key[j];
    {
        const double fx = (qx - g.minx) - (double)cx * g.h, fy = (qy - g.miny) - (double)cy * g.h, fz = (qz - g.minz) - (double)cz * g.h;
        double margin = __builtin_inf();
        const double eps = 1e-6 * g.h;
        if (cx - 1 > 0) margin = fmin(margin, fx + g.h - eps);
        if (cx + 1 < g.nx - 1) margin = fmin(margin, 2.0 * g.h - fx - eps);
        if (cy - 1 > 0) margin = fmin(margin, fy + g.h - eps);
        if (cy + 1 < g.ny - 1) margin = fmin(margin, 2.0 * g.h - fy - eps);
        if (cz - 1 > 0) margin = fmin(margin, fz + g.h - eps);
        if (cz + 1 < g.nz - 1) margin = fmin(margin, 2.0 * g.h - fz - eps);
        if (!(dk < margin * margin)) fb = true;
    }
    fb = fb && valid;
    {   // the uncertified queries go to the list of knn_listed_kernel (one counter update per wave)
        const unsigned long long m = __ballot(fb);
        if (m != 0ULL) {
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(ra.fb_count, (int)__popcll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            if (fb) ra.fb_list[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = (int32_t)q;
        }
    }
    KR_TICK(6);
    if (POS) {  // transposed lists in position space: lane = position, 256 contiguous bytes per neighbour slot and wave
        if (valid && !fb) {
#pragma unroll
            for (int j = 0; j < KR_MAX_K; ++j)
                if (j < k) a.idx_out[(int64_t)j * a.n + q] = pay[j];
            if (a.nn1_out && k > 1) a.nn1_out[q] = key[1];
            if (ra.normals_out) {
                float px[KR_MAX_K], py[KR_MAX_K], pz[KR_MAX_K];
#pragma unroll
                for (int j = 0; j < KR_MAX_K; ++j) {
                    const float4 c = a.sorted[j < k ? pay[j] : pay[0]];
                    px[j] = c.x; py[j] = c.y; pz[j] = c.z;
                }
                double nx, ny, nz;
                pca_normal_regs<KR_MAX_K>(px, py, pz, k, nx, ny, nz);
                double *o = ra.normals_out + 3 * q;
                o[0] = nx; o[1] = ny; o[2] = nz;
            }
        }
        KR_TICK(7);
        return;
    }
    {   // neighbour rows: staged in LDS (the list's space) and written out two rows per instruction, 120 contiguous bytes each
        // at k = 30, instead of 64 scattered dwords per instruction
        unsigned int *stage = s_list[wave];
        int *qid_of = reinterpret_cast<int *>(s_tile[wave]);
        qid_of[lane] = valid && !fb ? __float_as_int(qp.w) : -1;
#pragma unroll
        for (int j = 0; j < KR_MAX_K; ++j)
            if (j < k) stage[lane * k + j] = (unsigned int)pay[j];
        const float inv_k = 1.0f / (float)k;
        for (int t = lane; t < 64 * k; t += 64) {
            const int r = (int)(((float)t + 0.5f) * inv_k), c = t - r * k;
            const int id = qid_of[r];
            if (id >= 0) a.idx_out[(int64_t)id * k + c] = (int32_t)stage[t];
        }
    }
    if (!valid || fb) return;
    const int64_t row = (int64_t)__float_as_int(qp.w) * k;
    if (a.d2_out) {
#pragma unroll
        for (int j = 0; j < KR_MAX_K; ++j)
            if (j < k) a.d2_out[row + j] = key[j];
    }
    if (a.nn1_out && k > 1) a.nn1_out[__float_as_int(qp.w)] = key[1];
    if (ra.normals_out) {
        float px[KR_MAX_K], py[KR_MAX_K], pz[KR_MAX_K];
#pragma unroll
        for (int j = 0; j < KR_MAX_K; ++j) {
            const int64_t id = j < k ? pay[j] : pay[0];
            px[j] = ra.xyz[3 * id]; py[j] = ra.xyz[3 * id + 1]; pz[j] = ra.xyz[3 * id + 2];
        }
        double nx, ny, nz;
        pca_normal_regs<KR_MAX_K>(px, py, pz, k, nx, ny, nz);
        double *o = ra.normals_out + 3 * (int64_t)__float_as_int(qp.w);
        o[0] = nx; o[1] = ny; o[2] = nz;
    }
    KR_TICK(7);
}

// ---- PCA normals (pca_estimate_normals.h:43-108, unit weights), one thread per point ----------------
#pragma clang fp contract(off)
__device__ __forceinline__ void pca_normal_row(const float *__restrict__ xyz, const int32_t *__restrict__ nb, int k, double *__restrict__ out) {
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
    out[0] = nx; out[1] = ny; out[2] = nz;
}
// (one lane per point; the workgroup's 256 index rows are brought in with coalesced loads and read from LDS -- straight from
//  the row-major lists the lanes of a wave would touch 64 cache lines per load, the same ones for each of the k steps.  Dynamic
//  LDS: 256 * (k | 1) words.)
__global__ __launch_bounds__(256) void normals_kernel(const float *__restrict__ xyz, int64_t n, const int32_t *__restrict__ knn, int k,
                                                      double *__restrict__ normals) {
    extern __shared__ int32_t nrm_rows[];
    const int tid = (int)threadIdx.x, stride = k | 1;
    const int64_t base = (int64_t)blockIdx.x * 256;
    const int np = n - base < 256 ? (int)(n - base) : 256;
    for (int t = tid; t < np * k; t += 256) nrm_rows[(t / k) * stride + t % k] = knn[base * k + t];
    __syncthreads();
    if (tid < np) pca_normal_row(xyz, nrm_rows + tid * stride, k, normals + 3 * (base + tid));
}
__global__ void normals_listed_kernel(const float *__restrict__ xyz, const float4 *__restrict__ q_sorted, const int32_t *__restrict__ list,
                                      const int32_t *__restrict__ count, const int32_t *__restrict__ knn, int k,
                                      double *__restrict__ normals) {
    const int m = *count;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < m; i += (int)(gridDim.x * blockDim.x)) {
        const int64_t id = __float_as_int(q_sorted[list[i]].w);
        pca_normal_row(xyz, knn + id * k, k, normals + 3 * id);
    }
}

// position mode: the same from the transposed lists and the sorted array
__global__ void normals_listed_pos_kernel(const float4 *__restrict__ sorted, int64_t n, const int32_t *__restrict__ list,
                                          const int32_t *__restrict__ count, const int32_t *__restrict__ knnT, int k,
                                          double *__restrict__ normals) {
#pragma clang fp contract(off)
    const int m = *count;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < m; i += (int)(gridDim.x * blockDim.x)) {
        const int64_t pos = list[i];
        float px[F4L_MAX_K], py[F4L_MAX_K], pz[F4L_MAX_K];
        for (int j = 0; j < k; ++j) {
            const float4 c = sorted[knnT[(int64_t)j * n + pos]];
            px[j] = c.x; py[j] = c.y; pz[j] = c.z;
        }
        // pca_normal_row on gathered coordinates (same arithmetic and order)
        double cx = 0.0, cy = 0.0, cz = 0.0, sum = 0.0;
        for (int j = 0; j < k; ++j) { cx += (double)px[j]; cy += (double)py[j]; cz += (double)pz[j]; sum += 1.0; }
        const double inv = 1.0 / sum;
        cx *= inv; cy *= inv; cz *= inv;
        double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, ws = 0;
        for (int j = 0; j < k; ++j) {
            const double x = (double)px[j] - cx, y = (double)py[j] - cy, z = (double)pz[j] - cz;
            a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
            ws += 1.0;
        }
        const double t = 1.0 / ws;
        a00 *= t; a01 *= t; a02 *= t; a11 *= t; a12 *= t; a22 *= t;
        const double q = (a00 + a11 + a22) / 3.0;
        double pq = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
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
        else { const double sc = 1.0 / norm; nx *= sc; ny *= sc; nz *= sc; }
        normals[3 * pos] = nx; normals[3 * pos + 1] = ny; normals[3 * pos + 2] = nz;
    }
}
// the sorted array {x, y, z, id} as packed coordinates and the caller's index of every position
__global__ void unpack_sorted_kernel(const float4 *__restrict__ sorted, int64_t n, float *__restrict__ xyz_p, int32_t *__restrict__ orig) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 c = sorted[i];
        xyz_p[3 * i] = c.x; xyz_p[3 * i + 1] = c.y; xyz_p[3 * i + 2] = c.z;
        orig[i] = __float_as_int(c.w);
    }
}

__global__ void iota_kernel(int32_t *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}
// Sort keys of f4l_labels_to_csr: the label itself, or K for a label outside [0, K) (an "unlabelled" -1, a label beyond the
// caller's count): those points sort behind every patch -- order[off[K] ..) -- and belong to none, as the histogram this path
// replaced skipped them.  (Sorting the raw labels over the low bits of K only would place them by their low bits.)
// `via` (f4l_labels_to_csr_via): row i takes the label of row via[i] of another cloud (a via outside [0, n_labels) = no patch).
__global__ void label_keys_kernel(const int32_t *__restrict__ labels, int64_t n, int64_t K, int32_t *__restrict__ keys,
                                  int32_t *__restrict__ iota, const int32_t *__restrict__ via = nullptr, int64_t n_labels = 0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t l;
        if (via) {
            const int32_t v = via[i];
            l = (v >= 0 && (int64_t)v < n_labels) ? labels[v] : -1;
        } else l = labels[i];
        keys[i] = (l >= 0 && (int64_t)l < K) ? l : (int32_t)K;
        iota[i] = (int32_t)i;
    }
}
// CSR offsets straight from the SORTED keys (labels 0 .. K - 1, then K = "no patch"): off[l] = first position whose key is >= l.  Position i (0 .. n) writes the
// offsets of the labels that begin there: those above its left neighbour's label up to its own (one, unless labels are
// skipped); position n those above the last label up to K.  No histogram, no atomics, no scan.
__global__ void label_bounds_kernel(const int32_t *__restrict__ sorted, int64_t n, int64_t K, int64_t *__restrict__ off) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t a = i == 0 ? -1 : (int64_t)sorted[i - 1];
        int64_t b = i == n ? K : (int64_t)sorted[i];
        a = a < -1 ? -1 : a;
        b = b > K ? K : b;
        for (int64_t l = a + 1; l <= b; ++l) off[l] = i;
    }
}

constexpr int BBOX_BLOCKS = 2048;  // partial boxes of bbox_kernel (256 workgroups -- one per CU -- read at a third of the bandwidth)
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
    int32_t *fb_list, *fb_count, *run_lo, *run_hi, *pcell;
    int32_t *dense;      // [2 n + 4] (cell_range); in use when has_dense
    bool has_dense;
    float *bbox_partial;
    void *prim_temp;
    size_t prim_bytes, total;
};

// `small_k`: the layout for the lane-per-query walk of k <= KS_MAX_K alone (nn_small_kernel: no candidate runs, no redo list) --
// f4l_epoch_join's two clouds; 80 bytes per point less.
static int knn_ws_layout(int64_t n, KnnWs &w, unsigned char *base, bool small_k = false) {
    size_t sort_b = 0, rle_b = 0, scan_b = 0;
    unsigned long long *k0 = nullptr;
    int32_t *i0 = nullptr;
    if (rocprim::radix_sort_pairs<KeySortConfig>(nullptr, sort_b, k0, k0, i0, i0, (size_t)n, 0, 64, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::run_length_encode(nullptr, rle_b, k0, (unsigned int)n, k0, i0, i0, 0, false) != hipSuccess) return F4L_EHIP;
    {
        size_t s32 = 0, r32 = 0;
        unsigned int *q0 = nullptr;
        if (rocprim::radix_sort_pairs<KeySortConfig>(nullptr, s32, q0, q0, i0, i0, (size_t)n, 0, 32, 0, false) != hipSuccess) return F4L_EHIP;
        if (rocprim::run_length_encode(nullptr, r32, q0, (unsigned int)n, q0, i0, i0, 0, false) != hipSuccess) return F4L_EHIP;
        sort_b = sort_b > s32 ? sort_b : s32;
        rle_b = rle_b > r32 ? rle_b : r32;
    }
    if (rocprim::exclusive_scan(nullptr, scan_b, i0, i0, 0, (size_t)n + 1, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    {
        size_t d_b = 0;
        if (rocprim::inclusive_scan(nullptr, d_b, i0, i0, 2 * (size_t)n + 4, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
        scan_b = scan_b > d_b ? scan_b : d_b;
    }
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
    w.fb_list = (int32_t *)carve(small_k ? 256 : (size_t)n * 4);
    w.fb_count = (int32_t *)carve(256);
    w.run_lo = (int32_t *)carve(small_k ? 256 : (size_t)n * 9 * 4);
    w.run_hi = (int32_t *)carve(small_k ? 256 : (size_t)n * 9 * 4);
    w.pcell = (int32_t *)carve(small_k ? 256 : (size_t)n * 4);
    w.dense = (int32_t *)carve((2 * (size_t)n + 4) * 4);
    w.has_dense = false;
    w.bbox_partial = (float *)carve(BBOX_BLOCKS * 6 * 4);
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
    const unsigned bb_grid = grid_for(n, 256, BBOX_BLOCKS);
    hipLaunchKernelGGL(bbox_kernel, dim3(bb_grid), dim3(256), 0, st, xyz, n, partial);
    F4L_LAUNCH_CHECK();
    float hb[BBOX_BLOCKS * 6];
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

// ---- the same sizing without the host: f4l_knn on a stream that must not be synchronised (a HIP graph capture; F4L_KNN_ASYNC) ----
// One thread finishes the bounding box and sets the cell size from the surface-density guess of knn_build_grid -- without the
// correction from the measured occupancy, which would need the count on the host: exact for any cell size, tuned for terrain (where
// the guess is within the accepted band); the grid is kept below 2^32 cells so that the keys sort as 32-bit words.
__global__ void grid_init_kernel(const float *__restrict__ partial, int nb, int64_t n, int k, DevGrid *dg) {
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
            if (partial[6 * b + d] < mn[d]) mn[d] = partial[6 * b + d];
            if (partial[6 * b + 3 + d] > mx[d]) mx[d] = partial[6 * b + 3 + d];
        }
    bool ok = true;
    for (int d = 0; d < 3; ++d) ok = ok && mx[d] >= mn[d] && isfinite(mn[d]) && isfinite(mx[d]);
    if (!ok) { for (int d = 0; d < 3; ++d) { mn[d] = 0.0; mx[d] = 0.0; } }  // (NaN / inf coordinates: one cell; the sizing with the host refuses them)
    const double ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    double e0 = ext[0], e1 = ext[1], e2 = ext[2], t;
    if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 > e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
    const double diag = sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    const double target = k / 2.0 < 4.0 ? 4.0 : k / 2.0;
    double h;
    if (diag == 0.0) h = 1.0;
    else {
        const double area = (e2 > 0 ? e2 : diag) * (e1 > 0 ? e1 : (e2 > 0 ? e2 : diag) * 1e-3);
        h = sqrt(target * area / (double)n);
        if (!(h > 0.0)) h = diag;
    }
    GridSpec g;
    for (int iter = 0; iter < 64; ++iter) {
        const double hmin = (e2 > 0 ? e2 : 1.0) / 1048000.0;
        if (h < hmin) h = hmin;
        g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
        g.h = h; g.inv_h = 1.0 / h; g.inv_hz = g.inv_h;
        g.nx = (int)(ext[0] / h) + 1; g.ny = (int)(ext[1] / h) + 1; g.nz = (int)(ext[2] / h) + 1;
        if (g.nz > 1 && 5 * g.nz < (g.nx < g.ny ? g.nx : g.ny)) { g.nz = 1; g.inv_hz = 0.0; }
        if ((double)g.nx * (double)g.ny * (double)g.nz < 4294967295.0) break;
        h *= 1.26;  // (a volume with so many cells of this size: larger cells until the keys fit 32 bits)
    }
    dg->g = g;
    dg->M = 0;
    dg->dense_ok = (double)g.nx * (double)g.ny * (double)g.nz <= 2.0 * (double)n + 2.0 ? 1 : 0;
}
__global__ void grid_finish_kernel(DevGrid *dg, const int32_t *__restrict__ n_cells) { dg->M = *n_cells; }

static int knn_build_grid_async(const float *xyz, int64_t n, int k, KnnWs &w, hipStream_t st, DevGrid *dg) {
    const unsigned bb_grid = grid_for(n, 256, BBOX_BLOCKS);
    hipLaunchKernelGGL(bbox_kernel, dim3(bb_grid), dim3(256), 0, st, xyz, n, w.bbox_partial);
    hipLaunchKernelGGL(grid_init_kernel, dim3(1), dim3(1), 0, st, w.bbox_partial, (int)bb_grid, n, k, dg);
    F4L_LAUNCH_CHECK();
    unsigned int *k32a = reinterpret_cast<unsigned int *>(w.keys_a), *k32b = reinterpret_cast<unsigned int *>(w.keys_b);
    unsigned int *u32 = k32b + n;  // (second half of the 64-bit buffer)
    hipLaunchKernelGGL(cell_key32_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, GridSpec(), k32a, w.ids_a, (const DevGrid *)dg);
    F4L_LAUNCH_CHECK();
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(w.prim_temp, tb, k32a, k32b, w.ids_a, w.ids_b, (size_t)n, 0, 32u, st, false));
    // (the scans below run over the arrays' whole capacity -- the host does not know how many cells are occupied: zeros behind them)
    F4L_HIP_CHECK(hipMemsetAsync(w.cell_counts, 0, ((size_t)n + 1) * 4, st));
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::run_length_encode(w.prim_temp, tb, k32b, (unsigned int)n, u32, w.cell_counts, w.n_cells, st, false));
    hipLaunchKernelGGL(widen_keys_kernel, dim3(256), dim3(256), 0, st, u32, w.n_cells, w.cell_keys);
    hipLaunchKernelGGL(grid_finish_kernel, dim3(1), dim3(1), 0, st, dg, (const int32_t *)w.n_cells);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim_temp, tb, w.cell_counts, w.cell_start, 0, (size_t)n + 1, rocprim::plus<int32_t>(), st, false));
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, w.ids_b, n, w.sorted);
    F4L_LAUNCH_CHECK();
    F4L_HIP_CHECK(hipMemsetAsync(w.dense, 0, (2 * (size_t)n + 4) * 4, st));
    hipLaunchKernelGGL(dense_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.cell_keys, w.cell_counts, 0, w.dense, (const DevGrid *)dg);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::inclusive_scan(w.prim_temp, tb, w.dense, w.dense, 2 * (size_t)n + 4, rocprim::plus<int32_t>(), st, false));
    w.has_dense = true;
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
        g.h = h; g.inv_h = 1.0 / h; g.inv_hz = g.inv_h;
        g.nx = (int)(ext[0] / h) + 1; g.ny = (int)(ext[1] / h) + 1; g.nz = (int)(ext[2] / h) + 1;
        // A cloud that is flat along z (terrain: the case this library is for) gets ONE layer of cells: with cubic cells a
        // row of cells at fixed (y, z) breaks wherever the surface leaves the layer, and the lane-per-query kernel, whose
        // waves take consecutive points of a row, would see candidate runs 1.8x longer than needed (measured).  The search
        // stays exact for any cell shape: z then never limits a block.
        if (g.nz > 1 && 5 * g.nz < (g.nx < g.ny ? g.nx : g.ny)) { g.nz = 1; g.inv_hz = 0.0; }
        const double ncell = (double)g.nx * (double)g.ny * (double)g.nz;
        int end_bit = 1;
        while (end_bit < 63 && (double)(1ULL << end_bit) < ncell) ++end_bit;
        size_t tb = w.prim_bytes;
        if (end_bit <= 32) {
            unsigned int *k32a = reinterpret_cast<unsigned int *>(w.keys_a), *k32b = reinterpret_cast<unsigned int *>(w.keys_b);
            unsigned int *u32 = k32b + n;  // (second half of the 64-bit buffer)
            hipLaunchKernelGGL(cell_key32_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, g, k32a, w.ids_a);
            F4L_LAUNCH_CHECK();
            F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(w.prim_temp, tb, k32a, k32b, w.ids_a, w.ids_b, (size_t)n, 0, (unsigned)end_bit, st, false));
            tb = w.prim_bytes;
            F4L_HIP_CHECK(rocprim::run_length_encode(w.prim_temp, tb, k32b, (unsigned int)n, u32, w.cell_counts, w.n_cells, st, false));
            hipLaunchKernelGGL(widen_keys_kernel, dim3(256), dim3(256), 0, st, u32, w.n_cells, w.cell_keys);
            F4L_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(cell_key_kernel, dim3(grid_for(n)), dim3(256), 0, st, xyz, n, g, w.keys_a, w.ids_a);
            F4L_LAUNCH_CHECK();
            F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(w.prim_temp, tb, w.keys_a, w.keys_b, w.ids_a, w.ids_b, (size_t)n, 0,
                                                    (unsigned)end_bit, st, false));
            tb = w.prim_bytes;
            F4L_HIP_CHECK(rocprim::run_length_encode(w.prim_temp, tb, w.keys_b, (unsigned int)n, w.cell_keys, w.cell_counts,
                                                     w.n_cells, st, false));
        }
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
    // 5. one word per cell of the whole grid while that stays within two words per point (terrain tiles: always)
    w.has_dense = false;
    {
        const double ncell = (double)g.nx * (double)g.ny * (double)g.nz;
        if (ncell <= 2.0 * (double)n + 2.0 && !getenv("F4L_KNN_NO_DENSE")) {
            const size_t nc = (size_t)ncell;
            F4L_HIP_CHECK(hipMemsetAsync(w.dense, 0, (nc + 1) * 4, st));
            hipLaunchKernelGGL(dense_scatter_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, w.cell_keys, w.cell_counts, M, w.dense);
            F4L_LAUNCH_CHECK();
            tb = w.prim_bytes;
            F4L_HIP_CHECK(rocprim::inclusive_scan(w.prim_temp, tb, w.dense, w.dense, nc + 1, rocprim::plus<int32_t>(), st, false));
            w.has_dense = true;
        }
    }
    return F4L_OK;
}
}  // namespace f4l

namespace f4l {
// normals of listed queries from their finished neighbour rows (the queries knn_listed_kernel redid)
__global__ void normals_listed_kernel(const float *__restrict__ xyz, const float4 *__restrict__ q_sorted, const int32_t *__restrict__ list,
                                      const int32_t *__restrict__ count, const int32_t *__restrict__ knn, int k,
                                      double *__restrict__ normals);

// `pos` (position mode, see KnnArgs): idx_out receives the TRANSPOSED lists in position space, normals_out / nn1_out rows in
// position order, and pos->xyz_p / pos->orig the cloud in that order and the caller's index of every position.  Only the
// lane-per-query search has the mode (k <= KR_MAX_K): F4L_EUNSUPPORTED otherwise, the caller then takes the ordinary path.
struct KnnPosOut { float *xyz_p; int32_t *orig; };
static int knn_self(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, double *normals_out, void *workspace,
                    size_t workspace_bytes, hipStream_t st, double *nn1_out = nullptr, const KnnPosOut *pos = nullptr) {
    if (!xyz || n <= 0 || k < 1 || k > n || !idx_out || !workspace) return F4L_EINVAL;
    if (pos && (k > KR_MAX_K || getenv("F4L_KNN_WAVE_PER_QUERY") || d2_out)) return F4L_EUNSUPPORTED;
    if (k > F4L_MAX_K || n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    KnnWs w;
    int rc = knn_ws_layout(n, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    GridSpec g = GridSpec();
    int M = 0;
    // The grid is sized with the host in the loop (two read-backs: bounding box, occupied cells -- and a cell size corrected by
    // the measured occupancy) unless the stream must not be synchronised: while it is being captured into a HIP graph, or on
    // request (F4L_KNN_ASYNC); then the device sizes it (knn_build_grid_async) and the kernels read grid and counts there.
    DevGrid *dg = nullptr;
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        const bool lanes_or_small = !getenv("F4L_KNN_WAVE_PER_QUERY") && k <= KR_MAX_K;
        if ((cap != hipStreamCaptureStatusNone || getenv("F4L_KNN_ASYNC")) && lanes_or_small)
            dg = reinterpret_cast<DevGrid *>(reinterpret_cast<unsigned char *>(w.n_cells) + 64);
        else if (cap != hipStreamCaptureStatusNone)
            return F4L_EUNSUPPORTED;  // (the wave-per-query search sizes its launch by the occupied cells)
    }
    if (dg) {
        rc = knn_build_grid_async(xyz, n, k, w, st, dg);
        M = (int)n;  // (an upper bound, to size launches by: the kernels take the count from the device)
    } else {
        rc = knn_build_grid(xyz, n, k, w, st, g, M);
    }
    if (rc != F4L_OK) return rc;
    KnnArgs a;
    a.sorted = w.sorted; a.cell_keys = w.cell_keys; a.cell_start = w.cell_start; a.M = M; a.n = n; a.k = k; a.g = g;
    a.dense = w.has_dense ? w.dense : nullptr;
    a.q_sorted = w.sorted; a.q_cell_keys = w.cell_keys; a.q_cell_start = w.cell_start; a.Mq = M;
    a.idx_out = idx_out; a.d2_out = d2_out; a.dg = dg; a.nn1_out = nn1_out;
    a.pos_mode = pos ? 1 : 0;
    if (k <= KS_MAX_K && !pos && !normals_out && !nn1_out && !getenv("F4L_KNN_WAVE_PER_QUERY") && !getenv("F4L_KNN_NO_SMALL")) {
        launch_nn_small(a, n, st);
        F4L_LAUNCH_CHECK();
        return F4L_OK;
    }
    const bool lanes = k <= KR_MAX_K && !getenv("F4L_KNN_WAVE_PER_QUERY");  // (switch: A/B timing, and the test that both agree)
    if (!lanes) {
        // one wave per occupied cell, its queries one after the other
        hipLaunchKernelGGL(knn_cells_kernel, dim3((unsigned)((M + KNN_NW - 1) / KNN_NW)), dim3(KNN_NW * 64), 0, st, a);
        F4L_LAUNCH_CHECK();
        if (normals_out) {
            {
                const size_t lds = (size_t)256 * (k | 1) * 4;
                if (lds > 48 * 1024) F4L_HIP_CHECK(hipFuncSetAttribute((const void *)normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, st, xyz, n, idx_out, k, normals_out);
            }
            F4L_LAUNCH_CHECK();
        }
        return F4L_OK;
    }
    // one lane per query; the queries it cannot certify are listed and redone one wave each
    F4L_HIP_CHECK(hipMemsetAsync(w.fb_count, 0, 4, st));
    hipLaunchKernelGGL(cell_runs_kernel, dim3((unsigned)((9 * (int64_t)M + 255) / 256)), dim3(256), 0, st, w.cell_keys, w.cell_start, M,
                       w.cell_keys, w.cell_start, M, a.dense, g, w.run_lo, w.run_hi, w.pcell, (const DevGrid *)dg);
    F4L_LAUNCH_CHECK();
    KnnLanesArgs ra;
    ra.run_lo = w.run_lo; ra.run_hi = w.run_hi; ra.pcell = w.pcell;
    ra.prof = nullptr;
#ifdef F4L_KNN_PROF
    ra.prof = (unsigned long long *)(w.fb_count + 8);  // (the counter's 256-byte slot has room: 8 phases)
    F4L_HIP_CHECK(hipMemsetAsync(w.fb_count + 8, 0, 64, st));
#endif
    ra.a = a; ra.edge_slack = (float)(1e-6 * g.h * g.h); ra.fb_list = w.fb_list; ra.fb_count = w.fb_count; ra.xyz = xyz; ra.normals_out = normals_out;
    {
        const float top = (float)(4.0 * g.h * g.h);  // d2 = (2 h)^2 falls into the last bin
        unsigned int bits;
        memcpy(&bits, &top, 4);
        ra.bin_base = (int)(bits >> 21) - (KR_NB - 1);
    }
    if (pos) hipLaunchKernelGGL(knn_lanes_kernel<true>, dim3((unsigned)((n + KR_NW * 64 - 1) / (KR_NW * 64))), dim3(KR_NW * 64), 0, st, ra);
    else hipLaunchKernelGGL(knn_lanes_kernel<false>, dim3((unsigned)((n + KR_NW * 64 - 1) / (KR_NW * 64))), dim3(KR_NW * 64), 0, st, ra);
    F4L_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_listed_kernel, dim3(2048), dim3(KNN_NW * 64), 0, st, a, w.fb_list, w.fb_count);
    F4L_LAUNCH_CHECK();
#ifdef F4L_KNN_PROF
    {
        unsigned long long hp[8];
        int fbc = 0;
        F4L_HIP_CHECK(hipMemcpyAsync(hp, w.fb_count + 8, 64, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipMemcpyAsync(&fbc, w.fb_count, 4, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipStreamSynchronize(st));
        const double waves = (double)((n + 63) / 64);
        fprintf(stderr, "[knn prof] n=%lld k=%d cells=%d redone=%d | cycles per wave: setup %.0f pass1 %.0f thr %.0f pass2 %.0f load %.0f sort %.0f ties+check %.0f out %.0f\n",
                (long long)n, k, M, fbc, hp[0] / waves, hp[1] / waves, hp[2] / waves, hp[3] / waves, hp[4] / waves, hp[5] / waves, hp[6] / waves, hp[7] / waves);
    }
#endif
    if (normals_out) {
        if (pos) hipLaunchKernelGGL(normals_listed_pos_kernel, dim3(256), dim3(256), 0, st, w.sorted, n, w.fb_list, w.fb_count, idx_out, k, normals_out);
        else hipLaunchKernelGGL(normals_listed_kernel, dim3(256), dim3(256), 0, st, xyz, w.sorted, w.fb_list, w.fb_count, idx_out, k, normals_out);
        F4L_LAUNCH_CHECK();
    }
    if (pos) {
        hipLaunchKernelGGL(unpack_sorted_kernel, dim3(grid_for(n)), dim3(256), 0, st, w.sorted, n, pos->xyz_p, pos->orig);
        F4L_LAUNCH_CHECK();
    }
    return F4L_OK;
}
// (the partition's entry: supervoxel_gpu.hip)
int knn_position_mode(const float *xyz, int64_t n, int k, int32_t *knnT_out, double *normals_p_out, double *nn1_p_out, float *xyz_p_out,
                      int32_t *orig_out, void *workspace, size_t workspace_bytes, void *stream) {
    const KnnPosOut pos{xyz_p_out, orig_out};
    return knn_self(xyz, n, k, knnT_out, nullptr, normals_p_out, workspace, workspace_bytes, (hipStream_t)stream, nn1_p_out, &pos);
}
}  // namespace f4l

// Synchronises `stream` (the bounding box and the occupied-cell count are read back to size the grid).
extern "C" int f4l_knn(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, void *workspace,
                       size_t workspace_bytes, void *stream) {
    return f4l::knn_self(xyz, n, k, idx_out, d2_out, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

// f4l_knn with the PCA normal of every point's neighbour list (f4l_normals) computed in the same kernel, while the
// neighbours are in registers: supervoxel.cpp:105-113 in one launch.
extern "C" int f4l_knn_normals(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, double *normals_out,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!normals_out) return F4L_EINVAL;
    return f4l::knn_self(xyz, n, k, idx_out, d2_out, normals_out, workspace, workspace_bytes, (hipStream_t)stream);
}

// f4l_knn_normals that also hands out the squared distance of every point to its nearest other point (slot 1 of its
// row): what `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754) takes the median of -- the
// neighbour search of the partition then serves the resolution estimate too, and the 2-NN pass over the same cloud goes.
extern "C" int f4l_knn_normals_nn1(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, double *normals_out,
                                   double *nn1_d2_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!normals_out || !nn1_d2_out || k < 2) return F4L_EINVAL;
    return f4l::knn_self(xyz, n, k, idx_out, d2_out, normals_out, workspace, workspace_bytes, (hipStream_t)stream, nn1_d2_out);
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
    F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(wq.prim_temp, tb, wq.keys_a, wq.keys_b, wq.ids_a, wq.ids_b, (size_t)m, 0,
                                            (unsigned)end_bit, st, false));
    if (k <= KS_MAX_K && !getenv("F4L_KNN_WAVE_PER_QUERY") && !getenv("F4L_KNN_NO_SMALL")) {
        // a handful of neighbours: one lane per query (in cell order, for the locality of neighbouring lanes' reads); the
        // queries' own cell table is not needed
        hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(m)), dim3(256), 0, st, queries, wq.ids_b, m, wq.sorted);
        F4L_LAUNCH_CHECK();
        KnnArgs a;
        a.sorted = w.sorted; a.cell_keys = w.cell_keys; a.cell_start = w.cell_start; a.M = M; a.n = n; a.k = k; a.g = g;
        a.dense = w.has_dense ? w.dense : nullptr;
        a.q_sorted = wq.sorted; a.q_cell_keys = nullptr; a.q_cell_start = nullptr; a.Mq = 0;
        a.idx_out = idx_out; a.d2_out = d2_out; a.dg = nullptr;
        launch_nn_small(a, m, st);
        F4L_LAUNCH_CHECK();
        return F4L_OK;
    }
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
    a.dense = w.has_dense ? w.dense : nullptr;
    a.q_sorted = wq.sorted; a.q_cell_keys = wq.cell_keys; a.q_cell_start = wq.cell_start; a.Mq = Mq;
    a.idx_out = idx_out; a.d2_out = d2_out; a.dg = nullptr;
    if (k > KR_MAX_K || getenv("F4L_KNN_WAVE_PER_QUERY")) {
        hipLaunchKernelGGL(knn_cells_kernel, dim3((unsigned)((Mq + KNN_NW - 1) / KNN_NW)), dim3(KNN_NW * 64), 0, st, a);
        F4L_LAUNCH_CHECK();
        return F4L_OK;
    }
    // one lane per query (see knn_lanes_kernel); the runs of every occupied QUERY cell are looked up in the cloud's cell table
    F4L_HIP_CHECK(hipMemsetAsync(wq.fb_count, 0, 4, st));
    hipLaunchKernelGGL(cell_runs_kernel, dim3((unsigned)((9 * (int64_t)Mq + 255) / 256)), dim3(256), 0, st, wq.cell_keys, wq.cell_start, Mq,
                       w.cell_keys, w.cell_start, M, a.dense, g, wq.run_lo, wq.run_hi, wq.pcell);
    F4L_LAUNCH_CHECK();
    KnnLanesArgs ra;
    a.n = m;  // the lane kernel's query count
    ra.a = a; ra.run_lo = wq.run_lo; ra.run_hi = wq.run_hi; ra.pcell = wq.pcell; ra.fb_list = wq.fb_list; ra.fb_count = wq.fb_count;
    ra.xyz = nullptr; ra.normals_out = nullptr; ra.prof = nullptr; ra.edge_slack = (float)(1e-6 * g.h * g.h);
    {
        const float top = (float)(4.0 * g.h * g.h);
        unsigned int bits;
        memcpy(&bits, &top, 4);
        ra.bin_base = (int)(bits >> 21) - (KR_NB - 1);
    }
    hipLaunchKernelGGL(knn_lanes_kernel<false>, dim3((unsigned)((m + KR_NW * 64 - 1) / (KR_NW * 64))), dim3(KR_NW * 64), 0, st, ra);
    F4L_LAUNCH_CHECK();
    a.n = n;
    hipLaunchKernelGGL(knn_listed_kernel, dim3(2048), dim3(KNN_NW * 64), 0, st, a, wq.fb_list, wq.fb_count);
    F4L_LAUNCH_CHECK();
    if (getenv("F4L_KNN_DEBUG")) {  // (synchronises: measurements only)
        int fbc = 0;
        F4L_HIP_CHECK(hipMemcpyAsync(&fbc, wq.fb_count, 4, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipStreamSynchronize(st));
        fprintf(stderr, "[nn_query] n %lld m %lld k %d: h %.4f grid %d x %d x %d, %d cells (%d with queries), %d queries to the wave-per-query search\n",
                (long long)n, (long long)m, k, g.h, g.nx, g.ny, g.nz, M, Mq, fbc);
    }
    return F4L_OK;
}

// ---- the two searches over the SECOND epoch of a tile, with one binning of it -------------------------------------------------
// `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754) wants every target point's distance to its nearest
// other target point; the patches of the second epoch (a target point joins the patch of its nearest source point: what stands in
// for the learned patch matches, pipeline.py) want every target point's nearest SOURCE point.  f4l_knn(tgt, 2) + f4l_nn_query(src,
// tgt, 1) bin the target cloud twice -- once in its own grid, once (64-bit keys) in the source's; the order only matters for the
// locality of neighbouring lanes, so here the target is binned ONCE, in its own grid, and both searches walk it in that order.
extern "C" size_t f4l_epoch_join_workspace_bytes(int64_t n, int64_t m) {
    if (n <= 0 || m <= 0) return 0;
    f4l::KnnWs w, wq;
    if (f4l::knn_ws_layout(n, w, nullptr, true) != F4L_OK || f4l::knn_ws_layout(m, wq, nullptr, true) != F4L_OK) return 0;
    return w.total + wq.total;
}

// Synchronises `stream` (both grids are sized with the host in the loop).  tgt_nn1_d2_out [m] may be null (no median wanted).
extern "C" int f4l_epoch_join(const float *src, int64_t n, const float *tgt, int64_t m, double *tgt_nn1_d2_out, int32_t *tgt_to_src_out,
                              void *workspace, size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (m == 0) return F4L_OK;
    if (!src || !tgt || n <= 0 || m < 0 || !tgt_to_src_out || !workspace || (tgt_nn1_d2_out && m < 2)) return F4L_EINVAL;
    if (n > 0x7fffffffLL || m > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    KnnWs w, wq;
    int rc = knn_ws_layout(n, w, (unsigned char *)workspace, true);
    if (rc != F4L_OK) return rc;
    rc = knn_ws_layout(m, wq, (unsigned char *)workspace + w.total, true);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total + wq.total) return F4L_EWORKSPACE;
    GridSpec g, gq;
    int M = 0, Mq = 0;
    rc = knn_build_grid(tgt, m, 2, wq, st, gq, Mq);  // (rejects NaN / inf coordinates)
    if (rc != F4L_OK) return rc;
    rc = knn_build_grid(src, n, 1, w, st, g, M);
    if (rc != F4L_OK) return rc;
    KnnArgs a;
    a.d2_out = nullptr; a.dg = nullptr; a.q_cell_keys = nullptr; a.q_cell_start = nullptr; a.Mq = 0; a.q_sorted = wq.sorted;
    if (tgt_nn1_d2_out) {  // the target cloud against itself, k = 2: slot 1 of every row
        a.sorted = wq.sorted; a.cell_keys = wq.cell_keys; a.cell_start = wq.cell_start; a.M = Mq; a.n = m; a.k = 2; a.g = gq;
        a.dense = wq.has_dense ? wq.dense : nullptr;
        a.idx_out = nullptr; a.nn1_out = tgt_nn1_d2_out;
        launch_nn_small(a, m, st);
        F4L_LAUNCH_CHECK();
    }
    a.sorted = w.sorted; a.cell_keys = w.cell_keys; a.cell_start = w.cell_start; a.M = M; a.n = n; a.k = 1; a.g = g;
    a.dense = w.has_dense ? w.dense : nullptr;
    a.idx_out = tgt_to_src_out; a.nn1_out = nullptr;
    launch_nn_small(a, m, st);
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
    F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(w.prim_temp, tb, w.keys_a, w.keys_b, w.ids_a, w.ids_b, (size_t)n, 0,
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
    if (k > F4L_MAX_K) return F4L_EUNSUPPORTED;
    const size_t lds = (size_t)256 * (k | 1) * 4;
    if (lds > 48 * 1024) F4L_HIP_CHECK(hipFuncSetAttribute((const void *)f4l::normals_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(f4l::normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, (hipStream_t)stream, xyz, n,
                       knn_idx, k, normals_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

// ---- labels -> CSR ---------------------------------------------------------------------------------
namespace f4l {
struct CsrWs {
    int32_t *keys_out, *iota, *keys_in;
    void *prim_temp;
    size_t prim_bytes, total;
};
static int csr_ws_layout(int64_t n, int64_t K, CsrWs &w, unsigned char *base) {
    size_t sort_b = 0, scan_b = 0;
    int32_t *i0 = nullptr;
    unsigned long long *u0 = nullptr;
    if (rocprim::radix_sort_pairs<KeySortConfig>(nullptr, sort_b, i0, i0, i0, i0, (size_t)n, 0, 32, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::inclusive_scan(nullptr, scan_b, u0, u0, (size_t)K + 1, rocprim::plus<unsigned long long>(), 0, false) != hipSuccess)
        return F4L_EHIP;
    const size_t prim = sort_b > scan_b ? sort_b : scan_b;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    w.keys_out = (int32_t *)carve((size_t)n * 4);
    w.iota = (int32_t *)carve((size_t)n * 4);
    w.keys_in = (int32_t *)carve((size_t)n * 4);
    w.prim_temp = carve(prim);
    w.prim_bytes = prim;
    w.total = o;
    return F4L_OK;
}
}  // namespace f4l

// ---- median (the last step of `_compute_median_resolution`) ---------------------------------------------------------------
namespace f4l {
__global__ void median_pick_kernel(const double *__restrict__ middle, int64_t n, double *__restrict__ out, int of_sqrt) {
    // numpy.median: mean of the two middle elements (they coincide for odd n); middle = the elements of rank (n - 1) / 2, n / 2
    // (of_sqrt: the values are squares and the median wanted is that of their roots -- the root is monotone, so the middle
    //  elements are the same; the mean of the pair is taken over the roots)
    const double a = of_sqrt ? sqrt(middle[0]) : middle[0], b = of_sqrt ? sqrt(middle[1]) : middle[1];
    out[0] = (n & 1) ? a : (a + b) * 0.5;
}
}  // namespace f4l

extern "C" size_t f4l_median_f64_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return f4l::select_workspace_bytes() + 256;
}
// The two middle order statistics by radix select (select.hip): six passes that read the values, no sort.
namespace f4l {
static int median_impl(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace, size_t workspace_bytes, void *stream,
                       int of_sqrt);
}
extern "C" int f4l_median_f64(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace, size_t workspace_bytes,
                              void *stream) {
    return f4l::median_impl(values, n, stride, median_out, workspace, workspace_bytes, stream, 0);
}
// numpy.median(numpy.sqrt(values)) without the pass that takes the roots: `_compute_median_resolution` wants the median DISTANCE and
// the searches hand out squared distances (values >= 0).
extern "C" int f4l_median_sqrt_f64(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    return f4l::median_impl(values, n, stride, median_out, workspace, workspace_bytes, stream, 1);
}
static int f4l::median_impl(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace, size_t workspace_bytes, void *stream,
                            int of_sqrt) {
    using namespace f4l;
    if (!values || n <= 0 || stride < 1 || !median_out || !workspace) return F4L_EINVAL;
    if (workspace_bytes < f4l_median_f64_workspace_bytes(n)) return F4L_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double *middle = (double *)((unsigned char *)workspace + select_workspace_bytes());
    const int64_t ranks[2] = {(n - 1) / 2, n / 2};
    const int rc = select_ranks_f64(values, n, stride, 2, ranks, middle, workspace, st);
    if (rc != F4L_OK) return rc;
    hipLaunchKernelGGL(median_pick_kernel, dim3(1), dim3(1), 0, st, (const double *)middle, n, median_out, of_sqrt);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

extern "C" size_t f4l_labels_to_csr_workspace_bytes(int64_t n, int64_t K) {
    if (n <= 0 || K <= 0) return 0;
    f4l::CsrWs w;
    if (f4l::csr_ws_layout(n, K, w, nullptr) != F4L_OK) return 0;
    return w.total;
}

namespace f4l {
static int labels_to_csr_impl(const int32_t *labels, int64_t n_labels, const int32_t *via, int64_t n, int64_t K, int32_t *order_out,
                              int64_t *off_out, void *workspace, size_t workspace_bytes, void *stream);
}
extern "C" int f4l_labels_to_csr(const int32_t *labels, int64_t n, int64_t K, int32_t *order_out, int64_t *off_out,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    return f4l::labels_to_csr_impl(labels, n, nullptr, n, K, order_out, off_out, workspace, workspace_bytes, stream);
}
// The same for a cloud whose rows take their labels from ANOTHER cloud's rows: row i of the m rows belongs to patch
// labels[via[i]] (the second epoch's points joining the patch of their nearest first-epoch point: `labels[nn]` followed by
// f4l_labels_to_csr, without materialising the gathered labels).  A `via` outside [0, n_labels) is "no patch".
extern "C" int f4l_labels_to_csr_via(const int32_t *labels, int64_t n_labels, const int32_t *via, int64_t m, int64_t K, int32_t *order_out,
                                     int64_t *off_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (m > 0 && (!via || n_labels <= 0)) return F4L_EINVAL;
    return f4l::labels_to_csr_impl(labels, n_labels, via, m, K, order_out, off_out, workspace, workspace_bytes, stream);
}
static int f4l::labels_to_csr_impl(const int32_t *labels, int64_t n_labels, const int32_t *via, int64_t n, int64_t K, int32_t *order_out,
                                   int64_t *off_out, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace f4l;
    if (n < 0 || K <= 0 || !off_out || (n > 0 && (!labels || !order_out || !workspace))) return F4L_EINVAL;
    if (n > 0x7fffffffLL || K >= 0x7fffffffLL) return F4L_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        F4L_HIP_CHECK(hipMemsetAsync(off_out, 0, ((size_t)K + 1) * 8, st));
        return F4L_OK;
    }
    CsrWs w;
    int rc = csr_ws_layout(n, K, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    hipLaunchKernelGGL(label_keys_kernel, dim3(grid_for(n)), dim3(256), 0, st, labels, n, K, w.keys_in, w.iota, via, n_labels);
    F4L_LAUNCH_CHECK();
    int end_bit = 1;
    while (end_bit < 31 && (1LL << end_bit) <= K) ++end_bit;  // the keys run 0 .. K inclusive
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_pairs<KeySortConfig>(w.prim_temp, tb, w.keys_in, w.keys_out, w.iota, order_out, (size_t)n, 0,
                                            (unsigned)end_bit, st, false));  // LSD radix sort is stable
    hipLaunchKernelGGL(label_bounds_kernel, dim3(grid_for(n + 1)), dim3(256), 0, st, (const int32_t *)w.keys_out, n, K, off_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
