// supervoxel_gpu.hip -- the boundary-preserving supervoxel segmentation entirely on the device, without any host
// round trip: the parallel variant of codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-248 (+ the count of
// grid_sample.h:31-75, DisjointSet disjoint_set.h:59-85, Median median.h:22-31).
//
// The reference's fusion (:117-176) visits the representatives one after the other and is order dependent, so a parallel
// run cannot be label-identical (f4l_supervoxel keeps the sequential replay for that).  This variant keeps the
// algorithm's structure and every criterion, and replaces only the visiting order:
//
//   K        number of occupied cells of the resolution grid (grid_sample.h:48-68): bounding box by atomics, cell keys,
//            device radix sort, count of distinct keys -- never leaves the device.
//   lambda0  max(DBL_EPSILON, upper median of every point's smallest neighbour metric) (:105-113): device sort.
//   fusion   for lambda = lambda0, 2 lambda0, 4 lambda0 ... (:117): SUB-ROUNDS of conflict-free fusion on the graph of
//            representatives (directed edges u -> v: "v is a neighbour of a member of u", the reference's `adjacents`):
//            every representative draws a coin per sub-round; a tails representative v may be absorbed by a heads
//            neighbour u when the reference's own criterion holds, sizes[v] * metric(u, v) < lambda (:146-149), and
//            proposes to the u of smallest metric (ties: smallest index).  Heads are never absorbed and tails never
//            absorb in the same sub-round, so all proposals can be applied at once (Link(v, u), sizes[u] += sizes[v]).
//            Exactly like the reference's `if (--number_of_supervoxels == n_supervoxels) break` (:160), a sub-round never
//            goes below K: when it holds more proposals than representatives to spare, only the best (smallest loss;
//            ties by index) are applied -- an exact radix select on the device.
//            Edges are re-pointed to the current representatives every sub-round, self loops dropped and parallel edges
//            merged (hash set on the device) once per lambda.
//   labels   label = Find (:179-182) by pointer jumping.
//   exchange the boundary refinement (:186-237) as iterated relaxation: every sweep gives each point the label of the
//            neighbour's representative that is strictly closer than its own (the minimum over its neighbour list,
//            what the reference's scan over `neighbors[i]` ends with), double buffered; sweeps repeat until nothing
//            changes: the fixed point of the reference's queue ("no point has a neighbour whose representative is
//            strictly closer").
//   relabel  0..K-1 in ascending order of the representative's index (:241-247).
//
// Nothing here synchronises the stream or copies to the host: loop bounds live in a device-side state block, every
// kernel runs on a fixed grid and reads its trip counts from that block, and the host enqueues a fixed schedule of
// launches (the rounds and sweeps clouds normally need; those past the target count return at once) followed by one
// persistent kernel that loops over whatever is left (segment_rest_kernel).
// Results are deterministic (no result depends on the order atomics land in, nor on how the passes are split).
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "sv_metric.h"

namespace f4l {
namespace svg {

constexpr int LAMBDA_ROUNDS = 56;  // lambda0 * 2^55 exceeds any size * metric of a 2^31-point cloud
constexpr int SUBROUNDS = 3;
constexpr int SWEEPS = 96;
// what the schedule of LAUNCHES covers when the rest can run in one persistent kernel (segment_rest_kernel): a 1 M-point
// terrain tile at the reference's resolutions needs 11-14 rounds and 9-15 sweeps
constexpr int SCHED_ROUNDS = 16, SCHED_SWEEPS = 16;
constexpr unsigned GRID = 2048, BLOCK = 256;
constexpr unsigned long long DEAD = ~0ULL;
constexpr size_t OFFER_WAVES = 65536;  // >= waves of any grid the candidate passes run on (2048 x 4; the rest kernel: <= 1024 x 16)

struct State {
    double lambda;
    unsigned long long tau_excl;  // proposals with key < tau_excl are applied
    unsigned int bb[6];           // bounding box as order-preserving unsigned images of the floats (min x,y,z, max x,y,z)
    int32_t live, K;              // representatives now / wanted
    int32_t n_edges, n_edges_new;
    int32_t n_prop, round;
    int32_t stalled;              // the graph of representatives has no edges left but live > K
    int32_t sweeps_done, sweep_on, changed, full_sweep;
    int32_t n_labels;
    int32_t hash_factor;          // parallel edges are merged through the hash set once there are more than this many edges per representative
    // grid barrier of segment_rest_kernel: BAR_GROUPS groups of workgroups, each with its own arrival counter and generation word
    // (a cache line apart), one more counter for the groups' last arrivals
    unsigned int bar_top, bar_pad[31];
    unsigned int bar[2 * 16][32];
    int32_t bar_timeout;              // a workgroup gave up waiting at the barrier (never expected: status bit 3)
};

__device__ __forceinline__ unsigned int f2ord(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ unsigned long long d2ord(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ double ord2d(unsigned long long o) {
    return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffULL) : ~o));
}
__device__ __forceinline__ bool heads(int32_t v, int32_t round) {
    unsigned int h = (unsigned int)v * 0x9E3779B1u ^ ((unsigned int)round + 1u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (h & 1u) != 0u;
}
// One atomicAdd per WORKGROUP for all lanes with `take` (same-address atomics are slow: ~10 ns each): returns this lane's
// slot (valid where take).  Every thread of the block must call it (two barriers inside).
__device__ __forceinline__ int32_t block_append(int32_t *counter, bool take, int32_t *s_cnt /* [waves + 1] */) {
    const unsigned long long m = __ballot(take);
    const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 0) s_cnt[wave] = (int32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
        for (int w = 0; w < nw; ++w) { const int32_t c = s_cnt[w]; s_cnt[w] = tot; tot += c; }
        s_cnt[nw] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    const int32_t base = s_cnt[nw] + s_cnt[wave];
    __syncthreads();  // (s_cnt is reused by the next call)
    const unsigned int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    return base + (int32_t)below;
}
#define SV_FOR(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)

// ---- K: occupied cells of the resolution grid ------------------------------------------------------------------------
struct GridBox { int given; float mn[3], mx[3]; };
__global__ void init_state_kernel(State *st, int32_t n, GridBox box, int32_t hash_factor) {
    st->hash_factor = hash_factor;
    st->lambda = 0.0; st->tau_excl = 0ULL;
    for (int d = 0; d < 3; ++d) {  // the grid's anchor and extent: the cloud's own bounding box unless the caller gave one
        st->bb[d] = box.given ? f2ord(box.mn[d]) : 0xffffffffu;
        st->bb[3 + d] = box.given ? f2ord(box.mx[d]) : 0u;
    }
    st->live = n; st->K = 0; st->n_edges = 0; st->n_edges_new = 0; st->n_prop = 0; st->round = 0; st->stalled = 0;
    st->sweeps_done = 0; st->sweep_on = 1; st->changed = 0; st->full_sweep = 1; st->n_labels = 0;
    st->bar_top = 0u; st->bar_timeout = 0;
    for (int i = 0; i < 2 * 16; ++i) st->bar[i][0] = 0u;
}
__global__ void bbox_kernel(const float *__restrict__ xyz, int64_t n, State *st) {
    unsigned int mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    SV_FOR(i, n) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned int o = f2ord(xyz[3 * i + d]);
            mn[d] = o < mn[d] ? o : mn[d];
            mx[d] = o > mx[d] ? o : mx[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int a = (unsigned int)__shfl_xor((int)mn[d], m, 64), b = (unsigned int)__shfl_xor((int)mx[d], m, 64);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b > mx[d] ? b : mx[d];
        }
    }
    // one atomic per workgroup and bound (same-address atomics serialise: a launch of 256 workgroups, see the call)
    __shared__ unsigned int part[16][6];
    const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wave][d] = mn[d]; part[wave][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = (int)threadIdx.x;
        unsigned int v = part[0][d];
        for (int w = 1; w < nw; ++w) v = d < 3 ? (part[w][d] < v ? part[w][d] : v) : (part[w][d] > v ? part[w][d] : v);
        if (d < 3) atomicMin(&st->bb[d], v);
        else atomicMax(&st->bb[d], v);
    }
}
// grid_sample.h:48-68: size = int(len / res + 1), cell = clamp(int((p - min) / res)), all in double
__global__ void grid_key_kernel(const float *__restrict__ xyz, int64_t n, double resolution, const State *st,
                                unsigned long long *__restrict__ keys) {
#pragma clang fp contract(off)
    double mn[3];
    int size[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        mn[d] = (double)ord2f(st->bb[d]);
        size[d] = (int)(((double)ord2f(st->bb[3 + d]) - mn[d]) / resolution + 1);
    }
    SV_FOR(i, n) {
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)(((double)xyz[3 * i + d] - mn[d]) / resolution);
            c[d] = c[d] < 0 ? 0 : (c[d] > size[d] - 1 ? size[d] - 1 : c[d]);
        }
        keys[i] = ((unsigned long long)c[0] * (unsigned long long)size[1] + (unsigned long long)c[1]) * (unsigned long long)size[2] +
                  (unsigned long long)c[2];
    }
}
__global__ void count_distinct_kernel(const unsigned long long *__restrict__ sorted, int64_t n, State *st) {
    int cnt = 0;
    SV_FOR(i, n) cnt += (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
    cnt = wave_sum(cnt);
    if (lane_id() == 0 && cnt) atomicAdd(&st->K, cnt);
}

// ---- neighbour lists, transposed ----------------------------------------------------------------------------------------
// The passes that walk a point's neighbour list with one lane per point (lambda0's minimum metric, every sweep of the
// exchange) would read the row-major lists [n][k] with a stride of k words between lanes: 64 cache lines per load
// instruction, the same lines again for each of the k steps, and an L1 that 32 waves thrash.  They read knnT[j * n + i]
// instead: neighbour j of 64 consecutive points is 256 contiguous bytes.  One tiled transpose through LDS, 0.08 ms per 1 M
// points (lambda0's pass 0.83 -> 0.4 ms, a full sweep 0.5 -> 0.3 ms).
__global__ __launch_bounds__(256) void knn_transpose_kernel(const int32_t *__restrict__ knn, int64_t n, int k, int32_t *__restrict__ knnT) {
    __shared__ int32_t tile[64 * 65];
    const int tid = (int)threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
        const int np = n - base < 64 ? (int)(n - base) : 64;
        for (int t = tid; t < np * k; t += 256) tile[(t / k) * 65 + t % k] = knn[base * k + t];  // (rows padded to 65 words)
        __syncthreads();
        for (int t = tid; t < 64 * k; t += 256) {
            const int j = t >> 6, p = t & 63;
            if (p < np) knnT[(int64_t)j * n + base + p] = tile[p * 65 + j];
        }
        __syncthreads();
    }
}

// True when the distance term of metric(a, b) alone is at least `best` (with room for every rounding, and for a normal term
// that rounds a hair below zero): the metric cannot be below `best`, and the normals need not be fetched to know it.
__device__ __forceinline__ bool sv_metric_at_least(const float *__restrict__ xyz, int64_t a, int64_t b, double resolution, double best) {
    const double t1 = (double)xyz[3 * a] - xyz[3 * b], t2 = (double)xyz[3 * a + 1] - xyz[3 * b + 1], t3 = (double)xyz[3 * a + 2] - xyz[3 * b + 2];
    const double c = 0.4 / resolution, bound = best * 1.000001 + 1e-15;
    return c * c * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound;
}

// ---- lambda0 -----------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ void min_metric_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ knnT,
                                  int64_t n, int k, double resolution, double *__restrict__ dis0) {
    SV_FOR(i, n) {
        double best = DBL_MAX;
        for (int j = 0; j < k; ++j) {
            const int64_t q = knnT[(int64_t)j * n + i];
            if (q != i && q >= 0 && !sv_metric_at_least(xyz, i, q, resolution, best)) {  // (a negative entry = "no neighbour here": a point outside the caller's slab)
                const double m = sv_metric(xyz, nrm, i, q, resolution);
                best = m < best ? m : best;
            }
        }
        dis0[i] = best;
    }
}
__global__ void start_kernel(State *st, const double *__restrict__ dis_sorted, int64_t n, int k) {
    const double med = dis_sorted[n / 2];  // median.h:27-30: nth_element at size / 2
    st->lambda = med > DBL_EPSILON ? med : DBL_EPSILON;
    st->n_edges = (int32_t)0;  // set by init_edges_kernel's grid
    (void)k;
}
__global__ void init_points_kernel(int64_t n, int32_t *__restrict__ parent, int32_t *__restrict__ size,
                                   unsigned long long *__restrict__ bestm, int32_t *__restrict__ bestu) {
    SV_FOR(i, n) { parent[i] = (int32_t)i; size[i] = 1; bestm[i] = ~0ULL; bestu[i] = 0x7fffffff; }
}
__global__ void init_edges_kernel(const int32_t *__restrict__ knn, int64_t n, int k, unsigned long long *__restrict__ edges, State *st) {
    const int64_t total = n * k;
    SV_FOR(e, total) {
        const int64_t i = e / k;
        const int32_t q = knn[e];
        edges[e] = (q == (int32_t)i || q < 0 || q >= n) ? DEAD : (((unsigned long long)i << 32) | (unsigned int)q);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n_edges = (int32_t)(total > 0x7fffffffLL ? 0x7fffffffLL : total);
}

__device__ __forceinline__ bool fusing(const State *st) { return st->live > st->K && !st->stalled; }

// ---- one sub-round of conflict-free fusion -------------------------------------------------------------------------
// cand:  re-point every edge to the current representatives; an edge u -> v with u heads, v tails and
//        sizes[v] * metric(u, v) < lambda offers u to v: bestm[v] = min metric (atomicMin on the ordered image of the double).
//        The offering edges are also written out -- every WAVE owns a contiguous chunk of the edge list and compacts its
//        offers to the front of the same chunk of `offers` (the other edge buffer, idle until the lambda's merge), count in
//        `offer_cnt`: no atomics, no barriers --
// cand2: so that the tie pass (among the offers of smallest metric the smallest u wins: bestu[v] = min u) reads the offers only,
//        a few per cent of the list, instead of walking all edges again and measuring a quarter of them a second time.
// Both passes must run on the same grid (the chunks are derived from it).
__device__ __forceinline__ void wave_chunk(int32_t ne, int64_t &start, int64_t &end, int &wave_id) {
    const int waves = (int)(gridDim.x * (blockDim.x >> 6));
    wave_id = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int64_t chunk = (((int64_t)ne + waves - 1) / waves + 63) & ~(int64_t)63;
    start = (int64_t)wave_id * chunk;
    end = start + chunk < (int64_t)ne ? start + chunk : (int64_t)ne;
}
__device__ __forceinline__ void cand_body(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, State *st,
                                          unsigned long long *__restrict__ edges, const int32_t *__restrict__ parent,
                                          const int32_t *__restrict__ size, unsigned long long *__restrict__ bestm,
                                          unsigned long long *__restrict__ offers, int32_t *__restrict__ offer_cnt, bool hop) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n_prop = 0;  // (the previous sub-round's apply has finished)
    const int32_t ne = st->n_edges, round = st->round;
    const double lambda = st->lambda;
    int64_t start, end;
    int wave_id;
    wave_chunk(ne, start, end, wave_id);
    int cnt = 0;
    for (int64_t base = start; base < end; base += 64) {  // (uniform per wave)
        const int64_t e = base + lane_id();
        bool offer = false;
        unsigned long long key = DEAD;
        if (e < end) {
            key = edges[e];
            if (key != DEAD) {
                int32_t u = (int32_t)(key >> 32), v = (int32_t)(key & 0xffffffffULL);
                if (hop) {  // (not in the first sub-round of a lambda: the list was re-pointed when it was merged)
                    const int32_t pu = parent[u], pv = parent[v];  // both were representatives at the last re-pointing: one hop reaches the current ones
                    if (pu != u || pv != v) {
                        u = pu; v = pv;
                        key = u == v ? DEAD : (((unsigned long long)(unsigned int)u << 32) | (unsigned int)v);
                        edges[e] = key;
                    }
                }
                if (key != DEAD && heads(u, round) && !heads(v, round)) {
                    const double sz = (double)size[v];
                    // the distance term of the metric alone may already reach lambda (with room for every rounding, and for
                    // a normal term that rounds a hair below zero): such an edge is rejected without fetching the normals
                    const double t1 = (double)xyz[3 * u] - xyz[3 * v], t2 = (double)xyz[3 * u + 1] - xyz[3 * v + 1],
                                 t3 = (double)xyz[3 * u + 2] - xyz[3 * v + 2];
                    const double c = sz * 0.4 / resolution, bound = lambda * 1.000001 + sz * 1e-15;
                    if (!(c * c * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound)) {
                        const double m = sv_metric(xyz, nrm, u, v, resolution);
                        if (lambda - sz * m > 0.0) {  // :147-149 `improvement > 0.0`
                            atomicMin(&bestm[v], d2ord(m));
                            offer = true;
                        }
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(offer);
        if (offer)
            offers[start + cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u))] = key;
        cnt += (int)__popcll(mask);
    }
    if (lane_id() == 0) offer_cnt[wave_id] = cnt;
}
__device__ __forceinline__ void cand2_body(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, const State *st,
                                           const unsigned long long *__restrict__ bestm, int32_t *__restrict__ bestu,
                                           const unsigned long long *__restrict__ offers, const int32_t *__restrict__ offer_cnt) {
    int64_t start, end;
    int wave_id;
    wave_chunk(st->n_edges, start, end, wave_id);
    const int cnt = start < end ? offer_cnt[wave_id] : 0;
    for (int j = lane_id(); j < cnt; j += 64) {
        const unsigned long long key = offers[start + j];
        const int32_t u = (int32_t)(key >> 32), v = (int32_t)(key & 0xffffffffULL);
        if (bestm[v] == d2ord(sv_metric(xyz, nrm, u, v, resolution))) atomicMin(&bestu[v], u);
    }
}
__global__ void cand_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, State *st,
                            unsigned long long *__restrict__ edges, const int32_t *__restrict__ parent,
                            const int32_t *__restrict__ size, unsigned long long *__restrict__ bestm,
                            unsigned long long *__restrict__ offers, int32_t *__restrict__ offer_cnt, bool hop) {
    if (!fusing(st)) return;
    cand_body(xyz, nrm, resolution, st, edges, parent, size, bestm, offers, offer_cnt, hop);
}
__global__ void cand2_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, const State *st,
                             const unsigned long long *__restrict__ bestm, int32_t *__restrict__ bestu,
                             const unsigned long long *__restrict__ offers, const int32_t *__restrict__ offer_cnt) {
    if (!fusing(st)) return;
    cand2_body(xyz, nrm, resolution, st, bestm, bestu, offers, offer_cnt);
}
// collect: every tails representative with an offer becomes a proposal (key = float image of the loss : index)
__device__ __forceinline__ void collect_body(State *st, int64_t n, const int32_t *__restrict__ size, unsigned long long *__restrict__ bestm,
                                             int32_t *__restrict__ bestu, unsigned long long *__restrict__ prop_key, int32_t *__restrict__ prop_u) {
    __shared__ int32_t s_cnt[17];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x; v0 < n; v0 += stride) {  // whole workgroups iterate together
        const int64_t v = v0 + threadIdx.x;
        const int32_t u = v < n ? bestu[v] : 0x7fffffff;
        const bool has = u != 0x7fffffff;
        const int32_t at = block_append(&st->n_prop, has, s_cnt);
        if (!has) continue;
        const double loss = (double)size[v] * ord2d(bestm[v]);
        prop_key[at] = ((unsigned long long)f2ord((float)loss) << 32) | (unsigned int)v;
        prop_u[at] = u;
        bestm[v] = ~0ULL;
        bestu[v] = 0x7fffffff;
    }
}
__global__ __launch_bounds__(1024) void collect_kernel(State *st, int64_t n, const int32_t *__restrict__ size, unsigned long long *__restrict__ bestm,
                               int32_t *__restrict__ bestu, unsigned long long *__restrict__ prop_key, int32_t *__restrict__ prop_u) {
    if (!fusing(st)) return;
    collect_body(st, n, size, bestm, bestu, prop_key, prop_u);
}
// select (one workgroup): all proposals when there are representatives to spare, else exactly the (live - K) smallest
// keys -- radix select, 8 passes of 8 bits over the proposal list.
__device__ __forceinline__ void select_body(State *st, const unsigned long long *__restrict__ prop_key) {  // (1024 threads)
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_rank;
    const int a = st->n_prop, budget = st->live - st->K;
    const int tid = (int)threadIdx.x;
    if (a <= budget) {
        if (tid == 0) { st->tau_excl = ~0ULL; st->round += 1; }
        return;
    }
    if (tid == 0) { s_prefix = 0ULL; s_rank = budget; }  // 1-based rank of the last key to keep
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i = tid; i < a; i += 1024) {
            const unsigned long long key = prop_key[i];
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int r = s_rank, b = 0;
            unsigned int cum = 0;
            for (; b < 256; ++b) {
                if (cum + hist[b] >= (unsigned int)r) break;
                cum += hist[b];
            }
            s_rank = r - (int)cum;
            s_prefix = (prefix << 8) | (unsigned long long)b;
        }
        __syncthreads();
    }
    if (tid == 0) { st->tau_excl = s_prefix + 1ULL; st->round += 1; }  // keys are unique (index in the low word)
}
__global__ __launch_bounds__(1024) void select_kernel(State *st, const unsigned long long *__restrict__ prop_key) {
    if (!fusing(st)) return;
    select_body(st, prop_key);
}
__device__ __forceinline__ void apply_body(State *st, const unsigned long long *__restrict__ prop_key, const int32_t *__restrict__ prop_u,
                                           int32_t *__restrict__ parent, int32_t *__restrict__ size) {
    const int a = st->n_prop;
    const unsigned long long tau = st->tau_excl;
    int dropped = 0;
    SV_FOR(i, a) {
        const unsigned long long key = prop_key[i];
        if (key >= tau) continue;
        const int32_t v = (int32_t)(key & 0xffffffffULL), u = prop_u[i];
        parent[v] = u;                  // Link(v, u), disjoint_set.h:77-85
        atomicAdd(&size[u], size[v]);   // sizes[i] += sizes[j], :153
        ++dropped;
    }
    dropped = wave_sum(dropped);
    if (lane_id() == 0 && dropped) atomicSub(&st->live, dropped);
}
__global__ void apply_kernel(State *st, const unsigned long long *__restrict__ prop_key, const int32_t *__restrict__ prop_u,
                             int32_t *__restrict__ parent, int32_t *__restrict__ size) {
    if (!fusing(st)) return;
    apply_body(st, prop_key, prop_u, parent, size);
}

// ---- once per lambda: flatten the forest, merge parallel edges, double lambda ---------------------------------------
__device__ __forceinline__ void flatten_body(int64_t n, int32_t *__restrict__ parent) {
    SV_FOR(i, n) {
        int32_t r = parent[i];
        while (parent[r] != r) r = parent[r];  // (roots are stable while this pass runs)
        if (r != parent[i]) parent[i] = r;
    }
}
__global__ void flatten_kernel(const State *st, int64_t n, int32_t *__restrict__ parent, bool always) {
    if (!always && !fusing(st)) return;
    flatten_body(n, parent);
}
__device__ __forceinline__ int64_t table_size(int32_t n_edges) { return n_edges < 512 ? 1024 : 2 * (int64_t)n_edges; }
// Parallel edges are merged through the hash set only once they dominate the list (more than 40 edges per representative);
// before that the pass just drops the self loops -- the list of a young forest holds few duplicates and 30 M random table
// accesses cost more than they save.
__device__ __forceinline__ bool use_hash(const State *st) { return (int64_t)st->n_edges > (int64_t)st->hash_factor * (int64_t)st->live; }
__device__ __forceinline__ void table_clear_body(const State *st, unsigned long long *__restrict__ table) {
    if (!use_hash(st)) return;
    const int64_t ts = table_size(st->n_edges);
    SV_FOR(i, ts) table[i] = DEAD;
}
__global__ void table_clear_kernel(const State *st, unsigned long long *__restrict__ table) {
    if (!fusing(st)) return;
    table_clear_body(st, table);
}
__device__ __forceinline__ void dedup_body(State *st, const unsigned long long *__restrict__ edges, const int32_t *__restrict__ parent,
                                           unsigned long long *__restrict__ table, unsigned long long *__restrict__ edges_out) {
    const int32_t ne = st->n_edges;
    const unsigned long long ts = (unsigned long long)table_size(ne);
    __shared__ int32_t s_cnt[17];
    const bool hashed = use_hash(st);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x; e0 < ne; e0 += stride) {  // whole workgroups iterate together
        const int64_t e = e0 + threadIdx.x;
        const unsigned long long old = e < ne ? edges[e] : DEAD;
        bool fresh = false;
        unsigned long long key = DEAD;
        if (old != DEAD) {
            const int32_t u = parent[(int32_t)(old >> 32)], v = parent[(int32_t)(old & 0xffffffffULL)];
            if (u != v) {
                key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)v;
                fresh = !hashed;  // compaction only: every edge between two representatives is kept
                unsigned long long h = key * 0x9E3779B97F4A7C15ULL;
                h ^= h >> 29;
                unsigned long long slot = h % ts;
                while (hashed) {
                    unsigned long long seen = __atomic_load_n(&table[slot], __ATOMIC_RELAXED);  // most parallel edges stop here
                    if (seen == DEAD) seen = atomicCAS(&table[slot], DEAD, key);
                    if (seen == DEAD) { fresh = true; break; }  // first of its kind
                    if (seen == key) break;
                    slot = slot + 1 == ts ? 0 : slot + 1;
                }
            }
        }
        const int32_t at = block_append(&st->n_edges_new, fresh, s_cnt);  // one counter update per workgroup, not per edge
        if (fresh) edges_out[at] = key;
    }
}
__global__ __launch_bounds__(1024) void dedup_kernel(State *st, const unsigned long long *__restrict__ edges, const int32_t *__restrict__ parent,
                             unsigned long long *__restrict__ table, unsigned long long *__restrict__ edges_out) {
    if (!fusing(st)) return;
    dedup_body(st, edges, parent, table, edges_out);
}
__device__ __forceinline__ void next_lambda_body(State *st) {
    st->n_edges = st->n_edges_new;
    st->n_edges_new = 0;
    st->lambda *= 2.0;  // :117
    if (st->n_edges == 0) st->stalled = 1;  // disconnected graph of representatives: the reference would never return
}
__global__ void next_lambda_kernel(State *st) {
    if (!fusing(st)) return;
    next_lambda_body(st);
}

// ---- labels, boundary exchange, relabel ------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ void labels_init_body(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, int64_t n,
                                                 const int32_t *__restrict__ parent, int32_t *__restrict__ la, int32_t *__restrict__ lb,
                                                 double *__restrict__ dis) {
    SV_FOR(i, n) {
        const int32_t r = parent[i];
        la[i] = r;
        lb[i] = r;
        dis[i] = sv_metric(xyz, nrm, i, (int64_t)r, resolution);  // :186-189
    }
}
__global__ void labels_init_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, int64_t n,
                                   const int32_t *__restrict__ parent, int32_t *__restrict__ la, int32_t *__restrict__ lb,
                                   double *__restrict__ dis) {
    labels_init_body(xyz, nrm, resolution, n, parent, la, lb, dis);
}
// One sweep of the exchange (:214-226 for every point at once, reading the labels of the previous sweep).  A point is
// looked at when it, or a point that lists it or that it lists, changed in the previous sweep (`dirty`); `full_sweep`
// looks at every point (the first sweep, and the verification sweep that ends the relaxation).
__device__ __forceinline__ void sweep_body(const float *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ knnT,
                                           double resolution, int64_t n, int k, State *st, int32_t *__restrict__ l0, int32_t *__restrict__ l1,
                                           double *__restrict__ dis, unsigned char *__restrict__ d0, unsigned char *__restrict__ d1) {
    const bool odd = (st->sweeps_done & 1) != 0, full = st->full_sweep != 0;
    const int32_t *__restrict__ lin = odd ? l1 : l0;
    int32_t *__restrict__ lout = odd ? l0 : l1;
    // (a point reads only its OWN flag of the previous sweep and clears it on the way: the buffer is clean again when the
    // next sweep writes its flags into it)
    unsigned char *__restrict__ din = odd ? d1 : d0;
    unsigned char *__restrict__ dout = odd ? d0 : d1;
    bool any = false;
    SV_FOR(i, n) {
        const int32_t a = lin[i];
        int32_t bl = a;
        const bool look = full || din[i] != 0;
        din[i] = 0;
        if (look) {
            double best = dis[i];
            // (several neighbours carry the same foreign label: the last two labels found no better are not measured again --
            //  the best only decreases, so a label that lost once has lost for good)
            int32_t r0 = a, r1 = a;
            for (int j = 0; j < k; ++j) {
                const int32_t q = knnT[(int64_t)j * n + i];
                const int32_t b = q >= 0 ? lin[q] : a;
                if (b == a || b == bl || b == r0 || b == r1) continue;
                r1 = r0;
                if (sv_metric_at_least(xyz, i, (int64_t)b, resolution, best)) { r0 = b; continue; }
                const double d = sv_metric(xyz, nrm, i, (int64_t)b, resolution);
                if (d < best) { r0 = bl; best = d; bl = b; }
                else r0 = b;
            }
            if (bl != a) {
                dis[i] = best;
                any = true;
                dout[i] = 1;  // looked at again next sweep, together with the points it lists (:228-236)
                for (int j = 0; j < k; ++j) {
                    const int32_t q = knnT[(int64_t)j * n + i];
                    if (q >= 0) dout[q] = 1;
                }
            }
        }
        lout[i] = bl;
    }
    if (__ballot(any) != 0ULL && lane_id() == 0) atomicOr(&st->changed, 1);
}
__global__ void sweep_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ knnT,
                             double resolution, int64_t n, int k, State *st, int32_t *__restrict__ l0, int32_t *__restrict__ l1,
                             double *__restrict__ dis, unsigned char *__restrict__ d0, unsigned char *__restrict__ d1) {
    if (!st->sweep_on) return;
    sweep_body(xyz, nrm, knnT, resolution, n, k, st, l0, l1, dis, d0, d1);
}
__device__ __forceinline__ void sweep_end_body(State *st) {
    st->sweeps_done += 1;
    if (st->changed) { st->changed = 0; st->full_sweep = 0; }
    else if (!st->full_sweep) st->full_sweep = 1;                        // nothing left on the dirty lists: verify with a full sweep
    else st->sweep_on = 0;                                               // a full sweep changed nothing: fixed point
}
__global__ void sweep_end_kernel(State *st) {
    if (!st->sweep_on) return;
    sweep_end_body(st);
}
__global__ void root_flag_kernel(int64_t n, const int32_t *__restrict__ parent, int32_t *__restrict__ flag) {
    SV_FOR(i, n) flag[i] = parent[i] == (int32_t)i ? 1 : 0;
}
__global__ void relabel_kernel(State *st, int64_t n, const int32_t *__restrict__ l0, const int32_t *__restrict__ l1,
                               const int32_t *__restrict__ flag, const int32_t *__restrict__ rank, int32_t *__restrict__ labels_out,
                               int32_t *__restrict__ reps_out, int32_t *__restrict__ info_out) {
    const int32_t *__restrict__ lab = (st->sweeps_done & 1) ? l1 : l0;
    SV_FOR(i, n) {
        labels_out[i] = rank[lab[i]];  // :241-247: position of the representative in ascending index order
        if (reps_out && flag[i]) reps_out[rank[i]] = (int32_t)i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && info_out) {
        info_out[0] = rank[n - 1] + flag[n - 1];  // supervoxels produced
        info_out[1] = st->K;                      // occupied grid cells (the target)
        info_out[2] = (st->stalled ? 1 : 0) | (st->live > st->K && !st->stalled ? 2 : 0) | (st->sweep_on ? 4 : 0) | (st->bar_timeout ? 8 : 0);
        info_out[3] = st->sweeps_done;
    }
}


// ---- the rest of the fusion / of the exchange as ONE persistent launch ----------------------------------------------------
// The host cannot know how many lambda rounds and sweeps a cloud needs (no synchronisation), and a schedule long enough for
// every cloud (56 rounds, 96 sweeps: ~1350 launches) is mostly launches that return at once -- 1.4 us each, 1.4 of the
// 13.6 ms of a 1 M-point tile, which needs 14 rounds and 9 sweeps.  So the schedule of launches covers what clouds
// normally need (SCHED_ROUNDS, SCHED_SWEEPS), and whatever is left after it runs inside ONE kernel of as many workgroups as
// the chip holds at once: the same passes (the same device functions) separated by grid-wide barriers, in loops that END
// when the state says so.  Every decision to leave a loop reads state that was last written before the preceding barrier:
// all workgroups take the same branch and meet at the same barriers.  A barrier costs ~12 us against ~3 us for a kernel
// boundary (measured: the whole segmentation inside this kernel takes 14.4 ms, as launches 13.6 ms), which is why the
// passes that normally DO run stay launches.  The kernel is an ordinary launch on the caller's stream with the grid the
// occupancy query allows (all workgroups resident at once on an otherwise idle device; the barrier's wait is bounded in any
// case): hipLaunchCooperativeKernel goes through a queue of its own and cost 5 ms per call inside bench.py's process.
struct SegArgs {
    const float *xyz;
    const double *nrm;
    const int32_t *knnT;  // neighbour lists, transposed
    int64_t n;
    int k;
    double resolution;
    State *st;
    unsigned long long *edges_a, *edges_b, *table, *bestm, *prop_key;
    int32_t *parent, *size, *bestu, *prop_u, *la, *lb, *offer_cnt;
    double *dis;
    unsigned char *d0, *d1;
    int first_round;  // fusion: lambda rounds first_round .. LAMBDA_ROUNDS - 1 (the launches did the others); < 0: the sweeps
    int first_sweep;
};
// Grid-wide barrier between two passes.  Every wave first waits until its own stores have reached the L2 (vmcnt(0)); after
// the workgroup barrier ONE thread per workgroup writes the L2 back (release at agent scope), arrives, waits for the
// generation to change, and invalidates the caches (acquire at agent scope) -- 512 write-backs per barrier instead of one
// per wave, and the workgroups poll 16 words a cache line apart instead of one (cooperative_groups' grid.sync(): 45 us per
// barrier, 25 ms instead of 14 for the 1 M tile).  The wait is
// bounded (about a second): a workgroup that gives up sets bar_timeout and everything after runs to its end without
// waiting, so that the grid always drains.
constexpr unsigned int BAR_GROUPS = 16;
struct GridBarrier {
    State *st;
    unsigned int nblocks;
    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0 && !__hip_atomic_load(&st->bar_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned int g = blockIdx.x % BAR_GROUPS;
            const unsigned int members = nblocks / BAR_GROUPS + (g < nblocks % BAR_GROUPS ? 1u : 0u);
            const unsigned int groups = nblocks < BAR_GROUPS ? nblocks : BAR_GROUPS;
            unsigned int *cnt = &st->bar[2 * g][0], *gen_w = &st->bar[2 * g + 1][0];
            const unsigned int gen = __hip_atomic_load(gen_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool wait = true;
            if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(&st->bar_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u) {
                    __hip_atomic_store(&st->bar_top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (unsigned int q = 0; q < groups; ++q)
                        __hip_atomic_store(&st->bar[2 * q + 1][0], gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    wait = false;
                }
            }
            if (wait) {
                int spins = 0;
                while (__hip_atomic_load(gen_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) { __hip_atomic_store(&st->bar_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
};
__global__ __launch_bounds__(1024, 8) void segment_rest_kernel(SegArgs a) {
    State *st = a.st;
    const GridBarrier grid{st, gridDim.x};
    if (a.first_round >= 0) {
        // (an even number of rounds was scheduled: the edge list is back in edges_a)
        unsigned long long *cur = (a.first_round & 1) ? a.edges_b : a.edges_a, *nxt = (a.first_round & 1) ? a.edges_a : a.edges_b;
        for (int r = a.first_round; r < LAMBDA_ROUNDS && fusing(st); ++r) {
            for (int s = 0; s < SUBROUNDS && fusing(st); ++s) {
                cand_body(a.xyz, a.nrm, a.resolution, st, cur, a.parent, a.size, a.bestm, nxt, a.offer_cnt, s > 0);
                grid.sync();
                cand2_body(a.xyz, a.nrm, a.resolution, st, a.bestm, a.bestu, nxt, a.offer_cnt);
                grid.sync();
                collect_body(st, a.n, a.size, a.bestm, a.bestu, a.prop_key, a.prop_u);
                grid.sync();
                if (blockIdx.x == 0) select_body(st, a.prop_key);
                grid.sync();
                apply_body(st, a.prop_key, a.prop_u, a.parent, a.size);
                grid.sync();
            }
            if (!fusing(st)) break;
            flatten_body(a.n, a.parent);
            table_clear_body(st, a.table);  // (independent of the flattening: no barrier between them)
            grid.sync();
            dedup_body(st, cur, a.parent, a.table, nxt);
            grid.sync();
            if (blockIdx.x == 0 && threadIdx.x == 0) next_lambda_body(st);
            grid.sync();
            unsigned long long *t = cur; cur = nxt; nxt = t;
        }
    } else {
        for (int s = a.first_sweep; s < SWEEPS && st->sweep_on; ++s) {
            sweep_body(a.xyz, a.nrm, a.knnT, a.resolution, a.n, a.k, st, a.la, a.lb, a.dis, a.d0, a.d1);
            grid.sync();
            if (blockIdx.x == 0 && threadIdx.x == 0) sweep_end_body(st);
            grid.sync();
        }
    }
}
// Workgroups of segment_rest_kernel the device holds at once (0: unknown -> the whole schedule as launches).
static int segment_grid() {
    static int cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (cached[dev]) return cached[dev] < 0 ? 0 : cached[dev];
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, segment_rest_kernel, 1024, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        cached[dev] = -1;
        return 0;
    }
    cached[dev] = (size_t)cus * per_cu * 16 <= OFFER_WAVES ? cus * per_cu : -1;  // (16 waves per workgroup: one offer count each)
    return cached[dev] < 0 ? 0 : cached[dev];
}

static inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }

struct Ws {
    State *st;
    unsigned long long *keys_a, *keys_b, *edges_a, *edges_b, *table, *bestm, *prop_key;
    double *dis, *dis_sorted;
    int32_t *parent, *size, *bestu, *prop_u, *la, *lb, *flag, *rank, *offer_cnt, *knnT;
    unsigned char *d0, *d1;
    void *prim;
    size_t prim_bytes, total;
};
static int layout(int64_t n, int k, Ws &w, unsigned char *base) {
    size_t sort_u = 0, sort_d = 0, scan_b = 0;
    unsigned long long *u0 = nullptr;
    double *f0 = nullptr;
    int32_t *i0 = nullptr;
    if (rocprim::radix_sort_keys(nullptr, sort_u, u0, u0, (size_t)n, 0, 64, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::radix_sort_keys(nullptr, sort_d, f0, f0, (size_t)n, 0, 64, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::exclusive_scan(nullptr, scan_b, i0, i0, 0, (size_t)n, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    size_t prim = sort_u > sort_d ? sort_u : sort_d;
    prim = prim > scan_b ? prim : scan_b;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    const size_t ne = (size_t)n * (size_t)k;
    w.st = (State *)carve(sizeof(State));
    w.edges_a = (unsigned long long *)carve(ne * 8);
    w.edges_b = (unsigned long long *)carve(ne * 8);
    w.table = (unsigned long long *)carve((ne < 512 ? 1024 : 2 * ne) * 8);
    // the grid keys and the sorted metrics are dead before the edge table is first used: they alias it
    w.keys_a = w.table;
    w.keys_b = w.table ? w.table + n : nullptr;
    w.dis_sorted = (double *)w.keys_a;
    w.bestm = (unsigned long long *)carve((size_t)n * 8);
    w.prop_key = (unsigned long long *)carve((size_t)n * 8);
    w.dis = (double *)carve((size_t)n * 8);
    w.parent = (int32_t *)carve((size_t)n * 4);
    w.size = (int32_t *)carve((size_t)n * 4);
    w.bestu = (int32_t *)carve((size_t)n * 4);
    w.prop_u = (int32_t *)carve((size_t)n * 4);
    w.la = (int32_t *)carve((size_t)n * 4);
    w.lb = (int32_t *)carve((size_t)n * 4);
    w.flag = (int32_t *)carve((size_t)n * 4);
    w.rank = (int32_t *)carve((size_t)n * 4);
    w.knnT = (int32_t *)carve(ne * 4);
    w.offer_cnt = (int32_t *)carve(OFFER_WAVES * 4);  // one count per wave of the candidate passes
    w.d0 = carve((size_t)n);
    w.d1 = carve((size_t)n);
    w.prim = carve(prim);
    w.prim_bytes = prim;
    w.total = o;
    return F4L_OK;
}
}  // namespace svg
}  // namespace f4l

extern "C" size_t f4l_supervoxel_segment_device_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    f4l::svg::Ws w;
    if (f4l::svg::layout(n, k, w, nullptr) != F4L_OK) return 0;
    return w.total;
}

// Enqueues the whole segmentation on `stream`; never synchronises, never touches host memory.
extern "C" int f4l_supervoxel_segment_device(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k,
                                             double resolution, const float *grid_bbox_host, int32_t *labels_out,
                                             int32_t *reps_out, int32_t *info_out, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    using namespace f4l;
    using namespace f4l::svg;
    if (!xyz || !normals || !knn || n <= 0 || k < 1 || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n > 0x7fffffffLL) return F4L_EUNSUPPORTED;  // (the transpose tile holds rows of up to 64 neighbours)
    GridBox box;
    box.given = grid_bbox_host ? 1 : 0;
    for (int d = 0; d < 3; ++d) {
        box.mn[d] = grid_bbox_host ? grid_bbox_host[d] : 0.f;
        box.mx[d] = grid_bbox_host ? grid_bbox_host[3 + d] : 0.f;
        if (grid_bbox_host && !(box.mx[d] >= box.mn[d])) return F4L_EINVAL;
    }
    if (n > 0x7fffffffLL || (double)n * (double)k > 2147483647.0) return F4L_EUNSUPPORTED;
    Ws w;
    int rc = layout(n, k, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(GRID), b(BLOCK), one(1);

    int hash_factor = 40;
    if (const char *e = getenv("F4L_SV_HASH_FACTOR")) { if (atoi(e) > 0) hash_factor = atoi(e); }
    hipLaunchKernelGGL(init_state_kernel, one, one, 0, st, w.st, (int32_t)n, box, (int32_t)hash_factor);
    // K
    if (!box.given) hipLaunchKernelGGL(svg::bbox_kernel, dim3(256), b, 0, st, xyz, n, w.st);
    hipLaunchKernelGGL(grid_key_kernel, g, b, 0, st, xyz, n, resolution, w.st, w.keys_a);
    F4L_LAUNCH_CHECK();
    size_t tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_keys(w.prim, tb, w.keys_a, w.keys_b, (size_t)n, 0, 64, st, false));
    hipLaunchKernelGGL(count_distinct_kernel, g, b, 0, st, w.keys_b, n, w.st);
    // lambda0
    hipLaunchKernelGGL(knn_transpose_kernel, g, b, 0, st, knn, n, k, w.knnT);
    hipLaunchKernelGGL(min_metric_kernel, g, b, 0, st, xyz, normals, w.knnT, n, k, resolution, w.dis);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::radix_sort_keys(w.prim, tb, w.dis, w.dis_sorted, (size_t)n, 0, 64, st, false));
    hipLaunchKernelGGL(start_kernel, one, one, 0, st, w.st, w.dis_sorted, n, k);
    hipLaunchKernelGGL(init_points_kernel, g, b, 0, st, n, w.parent, w.size, w.bestm, w.bestu);
    hipLaunchKernelGGL(init_edges_kernel, g, b, 0, st, knn, n, k, w.edges_a, w.st);
    F4L_LAUNCH_CHECK();
    // fusion, labels and the exchange: the schedule of launches clouds normally need, then one persistent kernel for
    // whatever is left (see segment_rest_kernel).  F4L_SV_LAUNCHES=1: the whole
    // schedule as launches, whose tail returns at once.  F4L_SV_SCHEDULED="rounds,sweeps" overrides the split (tests run
    // "2,1": nearly everything inside the persistent kernel).
    const int rest_grid = getenv("F4L_SV_LAUNCHES") ? 0 : segment_grid();
    int sched_rounds = rest_grid > 0 ? SCHED_ROUNDS : LAMBDA_ROUNDS, sched_sweeps = rest_grid > 0 ? SCHED_SWEEPS : SWEEPS;
    if (const char *e = getenv("F4L_SV_SCHEDULED")) {
        int a = 0, c = 0;
        if (rest_grid > 0 && sscanf(e, "%d,%d", &a, &c) == 2 && a >= 0 && a <= LAMBDA_ROUNDS && c >= 0 && c <= SWEEPS) { sched_rounds = a; sched_sweeps = c; }
    }
    SegArgs sa;
    sa.xyz = xyz; sa.nrm = normals; sa.knnT = w.knnT; sa.n = n; sa.k = k; sa.resolution = resolution; sa.st = w.st;
    sa.edges_a = w.edges_a; sa.edges_b = w.edges_b; sa.table = w.table; sa.bestm = w.bestm; sa.prop_key = w.prop_key;
    sa.parent = w.parent; sa.size = w.size; sa.bestu = w.bestu; sa.prop_u = w.prop_u; sa.la = w.la; sa.lb = w.lb;
    sa.dis = w.dis; sa.d0 = w.d0; sa.d1 = w.d1; sa.offer_cnt = w.offer_cnt;
    unsigned long long *cur = w.edges_a, *nxt = w.edges_b;
    for (int r = 0; r < sched_rounds; ++r) {
        for (int s = 0; s < SUBROUNDS; ++s) {
            hipLaunchKernelGGL(cand_kernel, g, b, 0, st, xyz, normals, resolution, w.st, cur, w.parent, w.size, w.bestm, nxt, w.offer_cnt, s > 0);
            hipLaunchKernelGGL(cand2_kernel, g, b, 0, st, xyz, normals, resolution, w.st, w.bestm, w.bestu, nxt, w.offer_cnt);
            hipLaunchKernelGGL(collect_kernel, g, dim3(1024), 0, st, w.st, n, w.size, w.bestm, w.bestu, w.prop_key, w.prop_u);
            hipLaunchKernelGGL(select_kernel, one, dim3(1024), 0, st, w.st, w.prop_key);
            hipLaunchKernelGGL(apply_kernel, g, b, 0, st, w.st, w.prop_key, w.prop_u, w.parent, w.size);
        }
        hipLaunchKernelGGL(flatten_kernel, g, b, 0, st, w.st, n, w.parent, false);
        hipLaunchKernelGGL(table_clear_kernel, g, b, 0, st, w.st, w.table);
        hipLaunchKernelGGL(dedup_kernel, g, dim3(1024), 0, st, w.st, cur, w.parent, w.table, nxt);
        hipLaunchKernelGGL(next_lambda_kernel, one, one, 0, st, w.st);
        F4L_LAUNCH_CHECK();
        unsigned long long *t = cur; cur = nxt; nxt = t;
    }
    if (sched_rounds < LAMBDA_ROUNDS) {
        sa.first_round = sched_rounds; sa.first_sweep = 0;
        hipLaunchKernelGGL(segment_rest_kernel, dim3((unsigned)rest_grid), dim3(1024), 0, st, sa);
    }
    // labels and the boundary exchange
    hipLaunchKernelGGL(flatten_kernel, g, b, 0, st, w.st, n, w.parent, true);
    hipLaunchKernelGGL(labels_init_kernel, g, b, 0, st, xyz, normals, resolution, n, w.parent, w.la, w.lb, w.dis);
    F4L_HIP_CHECK(hipMemsetAsync(w.d0, 0, (size_t)n, st));
    F4L_HIP_CHECK(hipMemsetAsync(w.d1, 0, (size_t)n, st));
    for (int s = 0; s < sched_sweeps; ++s) {
        hipLaunchKernelGGL(sweep_kernel, g, b, 0, st, xyz, normals, w.knnT, resolution, n, k, w.st, w.la, w.lb, w.dis, w.d0, w.d1);
        hipLaunchKernelGGL(sweep_end_kernel, one, one, 0, st, w.st);
    }
    if (sched_sweeps < SWEEPS) {
        sa.first_round = -1; sa.first_sweep = sched_sweeps;
        hipLaunchKernelGGL(segment_rest_kernel, dim3((unsigned)rest_grid), dim3(1024), 0, st, sa);
    }
    F4L_LAUNCH_CHECK();
    // relabel 0..K-1 in ascending order of the representative's index
    hipLaunchKernelGGL(root_flag_kernel, g, b, 0, st, n, w.parent, w.flag);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim, tb, w.flag, w.rank, 0, (size_t)n, rocprim::plus<int32_t>(), st, false));
    hipLaunchKernelGGL(relabel_kernel, g, b, 0, st, w.st, n, w.la, w.lb, w.flag, w.rank, labels_out, reps_out, info_out);
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}

extern "C" size_t f4l_supervoxel_parallel_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    const size_t a = f4l_knn_workspace_bytes(n, k), s = f4l_supervoxel_segment_device_workspace_bytes(n, k);
    const size_t idx = ((size_t)n * k * 4 + 255) / 256 * 256, nrm = ((size_t)n * 24 + 255) / 256 * 256;
    return (a > s ? a : s) + idx + nrm;  // the kNN workspace is dead when the segmentation starts
}

// kNN + normals + segmentation, all on the device.  f4l_knn synchronises `stream` once while it sizes its grid (bounding
// box and cell count are read back); nothing after that does.
extern "C" int f4l_supervoxel_parallel(const float *xyz, int64_t n, int k, double resolution, int32_t *labels_out,
                                       int32_t *reps_out, int32_t *info_out, int32_t *knn_out, double *normals_out,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    if (!xyz || n <= 0 || k < 1 || k >= n || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;  // supervoxel.cpp:100
    if (k > F4L_MAX_K || n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    if (workspace_bytes < f4l_supervoxel_parallel_workspace_bytes(n, k)) return F4L_EWORKSPACE;
    const size_t a = f4l_knn_workspace_bytes(n, k), s = f4l_supervoxel_segment_device_workspace_bytes(n, k);
    const size_t shared = a > s ? a : s;
    const size_t idx_b = ((size_t)n * k * 4 + 255) / 256 * 256;
    unsigned char *base = (unsigned char *)workspace;
    int32_t *idx = knn_out ? knn_out : (int32_t *)(base + shared);
    double *nrm = normals_out ? normals_out : (double *)(base + shared + idx_b);
    int rc = f4l_knn_normals(xyz, n, k, idx, nullptr, nrm, workspace, a, stream);
    if (rc != F4L_OK) return rc;
    return f4l_supervoxel_segment_device(xyz, nrm, idx, n, k, resolution, nullptr, labels_out, reps_out, info_out, workspace, s, stream);
}
