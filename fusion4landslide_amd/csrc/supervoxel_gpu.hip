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
//   fusion   for lambda = lambda0, 2 lambda0, 4 lambda0 ... (:117), round r:
//            BUILD   the round's list of directed edges u -> v between representatives ("v is a neighbour of a member of
//                    u", the reference's `adjacents`): the previous round's list (round 1: the neighbour lists themselves)
//                    re-pointed to the current representatives, self loops dropped, parallel edges merged by a direct-mapped
//                    filter -- ONE pass over the list per round.  The same pass picks the round's ACTIVE edges, those the
//                    criterion sizes[v] * metric(u, v) < lambda (:146-149) accepts with the size v has when the round starts
//                    (sizes only grow): the only ones it can accept during the round -- a few per cent of the list (distance
//                    term alone >= lambda / sizes[v]: rejected without fetching the normals).
//            3 SUB-ROUNDS of conflict-free fusion over the active edges whose two ends are still representatives:
//                    every representative draws a coin per sub-round (hash of index and sub-round number); a tails v may be
//                    absorbed by a heads neighbour u when the criterion holds, and proposes to the u of smallest metric
//                    (ties: smallest index).  Heads are never absorbed and tails never absorb in the same sub-round, so all
//                    proposals apply at once (Link(v, u), sizes[u] += sizes[v]).  The adjacency of an absorbed
//                    representative reaches its new representative with the NEXT round's list.
//            Exactly like the reference's `if (--number_of_supervoxels == n_supervoxels) break` (:160), a sub-round never
//            goes below K: when it holds more proposals than representatives to spare, only the best (smallest loss; ties
//            by index) are applied -- an exact radix select, which can happen once (it ends the fusion) and runs after the
//            schedule.
//   labels   label = Find (:179-182).
//   exchange the boundary refinement (:186-237) as iterated relaxation: every sweep gives each point the label of the
//            neighbour's representative that is strictly closer than its own (the minimum over its neighbour list,
//            what the reference's scan over `neighbors[i]` ends with), double buffered; sweeps repeat until nothing
//            changes: the fixed point of the reference's queue ("no point has a neighbour whose representative is
//            strictly closer").
//   relabel  0..K-1 in ascending order of the representative's index (:241-247).
//
// Nothing here synchronises the stream or copies to the host.  Every kernel of the schedule knows its round / sub-round /
// sweep NUMBER from the host (a kernel argument) and finds everything else -- list lengths, proposal counts, the sweeps'
// on / full flags -- in per-round slots of a device-side state block that the kernels before it completed: no kernel
// exists only to advance a counter.  The host enqueues the rounds and sweeps clouds normally need (those past the target
// count return at once) followed by ONE-workgroup kernels that loop over whatever is left (segment_rest_kernel): same
// device functions, workgroup barriers, no assumption about what else is resident on the device.
// Results are deterministic (no result depends on the order atomics land in, nor on how the passes are split).
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "select.h"
#include "sv_metric.h"

namespace f4l {
namespace svg {

constexpr int LAMBDA_ROUNDS = 56;  // lambda0 * 2^55 exceeds any size * metric of a 2^31-point cloud
constexpr int SUBROUNDS = 3;
constexpr int SWEEPS = 1024;  // (the reference sweeps until nothing changes; a k = 8 graph over supervoxels of 1600 points needed more than 96)
// what the schedule of LAUNCHES covers (the rest runs in segment_rest_kernel): a 1 M-point terrain tile at the reference's
// resolutions needs 11-14 rounds and 5-15 sweeps
constexpr int SCHED_ROUNDS = 16, SCHED_SWEEPS = 16;
#ifndef SVX_GRID
#define SVX_GRID 2048
#endif
constexpr unsigned GRID = SVX_GRID, BLOCK = 256, GRID_ACTIVE = 512, BUILD_BLOCK = 1024;
constexpr unsigned long long DEAD = ~0ULL;
// The first rounds work straight from the neighbour lists: while the representatives are a handful of points each, a list of
// their edges would be nearly as long as the neighbour lists (18 M of 30 M edges after round 0 of a 1 M-point tile), and writing
// it costs a filter access per edge.  Rounds 0 .. KNN_ROUNDS - 1 only pick their active edges (the neighbour lists re-pointed on
// the fly); round KNN_ROUNDS builds the first list, when three quarters of a point's 30 edges have become parallel to one another
// and fall to the filter a wave keeps in LDS.
#ifndef SVX_KNN_ROUNDS
#define SVX_KNN_ROUNDS 4
#endif
constexpr int KNN_ROUNDS = SVX_KNN_ROUNDS;
constexpr int WAVE_FILTER = 256;  // slots of a wave's own filter (2 KB of LDS)
constexpr int32_t NONE = 0x7fffffff;
#ifndef SVX_WIDE_ROUNDS
#define SVX_WIDE_ROUNDS 6
#endif

struct State {
    double lambda0;
    unsigned long long tau_excl;                 // (overflow path) proposals with key < tau_excl are applied
    unsigned long long ne[LAMBDA_ROUNDS + 1];    // ne[r]: edges of round r's list, r >= 1 (round 0's list is the neighbour lists)
    unsigned long long na[LAMBDA_ROUNDS + 1];    // na[r]: active edges of round r
    unsigned long long n_off[LAMBDA_ROUNDS * SUBROUNDS];  // offers of sub-round rho
    unsigned long long nrep[LAMBDA_ROUNDS + 2];  // nrep[r], r > KNN_ROUNDS: points whose record round r refreshes (node_body)
    unsigned long long n_list;                   // (overflow path) proposals collected
    unsigned int bb[6];                          // the box of the resolution grid as order-preserving unsigned images of the floats (min x,y,z, max x,y,z)
    unsigned int pb[6];                          // the points' own bounding box (the frame of the quantised positions)
    int32_t live, live_snap, K;                  // representatives now / when the current sub-round started / wanted
    int32_t overflow;                            // 1 + the sub-round whose proposals exceeded the representatives to spare (0: none)
    int32_t stalled;                             // the graph of representatives has no edges left but live > K
    int32_t rounds;                              // lambda rounds entered
    int32_t sweeps_done;
    int32_t overflow_offers;                     // a sub-round made more offers than its list holds (refused: status bit 3)
    int32_t unsorted;                            // some neighbour list is not in ascending order of distance (round 0 then reads every list to its end)
    int32_t n_prop[LAMBDA_ROUNDS * SUBROUNDS];   // proposals of sub-round rho
    int32_t sw_on[SWEEPS + 1], sw_full[SWEEPS + 1], sw_changed[SWEEPS + 1];  // per sweep: ran / looked at every point / changed a label
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
// An edge key is (u << 32) | v with u, v PLACES below 2^28 (f4l_supervoxel_segment_device refuses more points; its workspace for 2^28
// points is beyond one MI355X anyway).  The eight spare bits -- 60..63 and 28..31 -- carry
//   in the filter table: the number of the lambda round that wrote the entry: entries of earlier rounds never compare equal, and
//     the table is cleared ONCE per segmentation instead of once per round (round 4: 2 GB of zero fill per 10 M points);
//   in the list of active edges: bits 60..62 = the sub-rounds in which the edge can make an offer (u heads and v tails), known
//     when the edge is picked (the coins of a round's representatives travel in their records): a sub-round tests its bit before
//     it touches anything else, and an edge with no bit set is never listed.
constexpr unsigned long long PLACE_MASK = 0x0fffffffULL;
constexpr int64_t MAX_PLACES = 1LL << 28;
__device__ __forceinline__ int32_t key_u(unsigned long long key) { return (int32_t)((key >> 32) & PLACE_MASK); }
__device__ __forceinline__ int32_t key_v(unsigned long long key) { return (int32_t)(key & PLACE_MASK); }
__device__ __forceinline__ unsigned long long round_stamp(int r) {
    const unsigned long long e = (unsigned long long)(r + 1);  // 1 .. LAMBDA_ROUNDS: never the 0 of a cleared or plain key
    return ((e & 15ULL) << 60) | (((e >> 4) & 15ULL) << 28);
}
__device__ __forceinline__ unsigned int lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
// One atomicAdd per WORKGROUP and counter for all the workgroup's items with t0 / t1 (same-address atomics are slow: ~10 ns
// each, and they serialise: a counter touched once per wave by 4096 waves costs 40 us whatever else the kernel does): the
// slots of this lane's ITEMS items (valid where taken).  Every thread of the block must call it (three barriers inside).
struct AppendScratch { int32_t c0[17], c1[17]; unsigned long long base[2]; };
template <int ITEMS>
__device__ __forceinline__ void block_append2(unsigned long long *counter0, const bool (&t0)[ITEMS], unsigned long long *counter1,
                                              const bool (&t1)[ITEMS], unsigned long long (&at0)[ITEMS], unsigned long long (&at1)[ITEMS],
                                              AppendScratch &s) {
    unsigned long long m0[ITEMS], m1[ITEMS];
    int n0 = 0, n1 = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        m0[j] = __ballot(t0[j]);
        m1[j] = __ballot(t1[j]);
        n0 += (int)__popcll(m0[j]);
        n1 += (int)__popcll(m1[j]);
    }
    const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 0) { s.c0[wave] = n0; s.c1[wave] = n1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot0 = 0, tot1 = 0;
        for (int w = 0; w < nw; ++w) {
            const int32_t a = s.c0[w], b = s.c1[w];
            s.c0[w] = tot0; s.c1[w] = tot1;
            tot0 += a; tot1 += b;
        }
        s.base[0] = tot0 ? atomicAdd(counter0, (unsigned long long)tot0) : 0ULL;
        s.base[1] = tot1 ? atomicAdd(counter1, (unsigned long long)tot1) : 0ULL;
    }
    __syncthreads();
    unsigned long long b0 = s.base[0] + (unsigned long long)s.c0[wave], b1 = s.base[1] + (unsigned long long)s.c1[wave];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        at0[j] = b0 + lanes_below(m0[j]);
        at1[j] = b1 + lanes_below(m1[j]);
        b0 += (unsigned long long)__popcll(m0[j]);
        b1 += (unsigned long long)__popcll(m1[j]);
    }
    __syncthreads();  // (the scratch is reused by the next call)
}
// Sum of `v` over the workgroup added to *counter with ONE atomic (by thread 0; nothing when the sum is 0).  Every thread calls it.
__device__ __forceinline__ void block_add(int32_t *counter, int v, bool subtract) {
    __shared__ int part[16];
    v = wave_sum(v);
    if (lane_id() == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += part[w];
        if (tot) { if (subtract) atomicSub(counter, tot); else atomicAdd(counter, tot); }
    }
}
#define SV_FOR(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)

// ---- K: occupied cells of the resolution grid ------------------------------------------------------------------------
struct GridBox { int given; float mn[3], mx[3]; };
__global__ void init_state_kernel(State *st, int32_t n, GridBox box) {  // (the block was zeroed by a memset before)
    for (int d = 0; d < 3; ++d) {  // the grid's anchor and extent: the cloud's own bounding box unless the caller gave one
        st->bb[d] = box.given ? f2ord(box.mn[d]) : 0xffffffffu;
        st->bb[3 + d] = box.given ? f2ord(box.mx[d]) : 0u;
        st->pb[d] = 0xffffffffu;
        st->pb[3 + d] = 0u;
    }
    st->live = n;
    st->live_snap = n;
}
__global__ void bbox_kernel(const float *__restrict__ xyz, int64_t n, State *st, int grid_box_given) {
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
        if (d < 3) { atomicMin(&st->pb[d], v); if (!grid_box_given) atomicMin(&st->bb[d], v); }
        else { atomicMax(&st->pb[d], v); if (!grid_box_given) atomicMax(&st->bb[d], v); }
    }
}
// grid_sample.h:48-68: size = int(len / res + 1), cell = clamp(int((p - min) / res)), all in double.  K is the number of DISTINCT
// cell keys: counted by inserting the keys into an open-addressing hash set (linear probing, compare-and-swap; at least two
// slots per point, so a probe always ends) -- rounds 2-4 sorted the 64-bit keys for it, eight radix passes over the cloud.  The
// points arrive in the segmentation's spatial order, so most lanes hold the key of the lane before them and insert nothing.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__global__ void grid_count_kernel(const float *__restrict__ xyz, int64_t n, double resolution, State *st,
                                  unsigned long long *__restrict__ set, unsigned long long slots) {
#pragma clang fp contract(off)
    double mn[3];
    int size[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        mn[d] = (double)ord2f(st->bb[d]);
        size[d] = (int)(((double)ord2f(st->bb[3 + d]) - mn[d]) / resolution + 1);
    }
    int cnt = 0;
    SV_FOR(i, n) {
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)(((double)xyz[3 * i + d] - mn[d]) / resolution);
            c[d] = c[d] < 0 ? 0 : (c[d] > size[d] - 1 ? size[d] - 1 : c[d]);
        }
        const unsigned long long key = ((unsigned long long)c[0] * (unsigned long long)size[1] + (unsigned long long)c[1]) * (unsigned long long)size[2] +
                                       (unsigned long long)c[2];
        const unsigned long long before = ((unsigned long long)(unsigned int)__shfl_up((int)(key >> 32), 1, 64) << 32) |
                                          (unsigned int)__shfl_up((int)(key & 0xffffffffULL), 1, 64);
        if (lane_id() > 0 && before == key) continue;  // (the lane before inserts it)
        unsigned long long slot = __umul64hi(mix64(key), slots);
        for (;;) {
            const unsigned long long old = atomicCAS(&set[slot], DEAD, key);  // (keys are below 2^63: never DEAD)
            if (old == DEAD) { ++cnt; break; }
            if (old == key) break;
            slot = slot + 1ULL == slots ? 0ULL : slot + 1ULL;
        }
    }
    cnt = wave_sum(cnt);
    __shared__ int part[16];
    if (lane_id() == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per workgroup
        int tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += part[w];
        if (tot) atomicAdd(&st->K, tot);
    }
}

// ---- the segmentation's own order of the points ----------------------------------------------------------------------------
// Every pass gathers records of a point's neighbours; with the points in the caller's order (a scanner's, or no order at all)
// each gather is a cache line of its own from HBM: 10 KB of traffic per point in all.  The segmentation therefore works on a copy
// sorted along a Z-curve of (x, y) cells of half the resolution: neighbours share lines.  Everything that DECIDES by index --
// the coin of a sub-round, the smallest index among equal offers, the order of the labels -- uses the caller's index of a point
// (`orig`), so the result is the one the caller's order gives.
__device__ __forceinline__ unsigned int spread16(unsigned int v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__global__ void order_key_kernel(const float *__restrict__ xyz, int64_t n, double resolution, const State *st, int identity,
                                 unsigned int *__restrict__ keys, int32_t *__restrict__ ids) {
    const float x0 = ord2f(st->pb[0]), y0 = ord2f(st->pb[1]);
    const float ex = ord2f(st->pb[3]) - x0, ey = ord2f(st->pb[4]) - y0, e = ex > ey ? ex : ey;
    float cell = (float)(0.5 * resolution);
    if (e / cell > 65535.f) cell = e / 65535.f;
    const float inv = 1.f / cell;
    SV_FOR(i, n) {
        float cx = (xyz[3 * i] - x0) * inv, cy = (xyz[3 * i + 1] - y0) * inv;
        cx = cx < 0.f ? 0.f : (cx > 65535.f ? 65535.f : cx);
        cy = cy < 0.f ? 0.f : (cy > 65535.f ? 65535.f : cy);
        keys[i] = identity ? (unsigned int)i : (spread16((unsigned int)cx) | (spread16((unsigned int)cy) << 1));
        ids[i] = (int32_t)i;
    }
}
__global__ void order_apply_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ orig, int64_t n,
                                   float *__restrict__ xyz_p, double *__restrict__ nrm_p, int32_t *__restrict__ pos_of) {
    SV_FOR(i, n) {
        const int64_t o = orig[i];
        pos_of[o] = (int32_t)i;
#pragma unroll
        for (int d = 0; d < 3; ++d) { xyz_p[3 * i + d] = xyz[3 * o + d]; nrm_p[3 * i + d] = nrm[3 * o + d]; }
    }
}
// ---- neighbour lists, transposed ----------------------------------------------------------------------------------------
// The passes that walk a point's neighbour list with one lane per point (lambda0's minimum metric, every sweep of the
// exchange) would read the row-major lists [n][k] with a stride of k words between lanes: 64 cache lines per load
// instruction, the same lines again for each of the k steps, and an L1 that 32 waves thrash.  They read knnT[j * n + i]
// instead: neighbour j of 64 consecutive points is 256 contiguous bytes.  One tiled transpose through LDS, 0.08 ms per 1 M
// points (lambda0's pass 0.83 -> 0.4 ms, a full sweep 0.5 -> 0.3 ms).
__global__ __launch_bounds__(256) void knn_transpose_kernel(const int32_t *__restrict__ knn, const int32_t *__restrict__ orig,
                                                            const int32_t *__restrict__ pos_of, int64_t n, int k, int32_t *__restrict__ knnT) {
    __shared__ int32_t tile[64 * 65];
    const int tid = (int)threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
        const int np = n - base < 64 ? (int)(n - base) : 64;
        // the row of the point at place base + p is the caller's row orig[base + p]; its entries become places
        for (int t = tid; t < np * k; t += 256) {
            const int p = t / k, j = t % k;
            const int32_t q = knn[(int64_t)orig[base + p] * k + j];
            tile[p * 65 + j] = (q < 0 || (int64_t)q >= n) ? -1 : pos_of[q];  // (rows padded to 65 words)
        }
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

// c * |p_a - p_b| > bound, exactly as sv_metric_at_least evaluates it
__device__ __forceinline__ bool sv_metric_at_least_sized(const float *__restrict__ xyz, int64_t a, int64_t b, double c, double bound) {
    const double t1 = (double)xyz[3 * a] - xyz[3 * b], t2 = (double)xyz[3 * a + 1] - xyz[3 * b + 1], t3 = (double)xyz[3 * a + 2] - xyz[3 * b + 2];
    return c * c * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound;
}
__device__ __forceinline__ bool sv_distance_at_least(const float (&pa)[3], const float (&pb)[3], double resolution, double best) {
    const double t1 = (double)pa[0] - pb[0], t2 = (double)pa[1] - pb[1], t3 = (double)pa[2] - pb[2];
    const double c = 0.4 / resolution, bound = best * 1.000001 + 1e-15;
    return c * c * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound;
}

// ---- lambda0 -----------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// SORTED: the lists are known to be in ascending order of distance (the library's own search, position mode): the metric is at
// least its distance term, so the first neighbour whose distance term reaches the smallest metric so far ends the row -- the
// minimum sits among the first few neighbours, and a row reads a chunk or two of its list instead of all of it.  Positions AND
// normals of a chunk are requested together (one round trip per chunk).
template <bool SORTED>
__global__ void min_metric_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ knnT,
                                  int64_t n, int k, double resolution, double *__restrict__ dis0, State *st) {
    constexpr int CH = 6;  // neighbours whose indices and coordinates are loaded together (see rows_body)
    bool unsorted = false;  // (a list that is NOT in ascending order of distance: round 0 may then not stop at the first far neighbour)
    SV_FOR(i, n) {
        double best = DBL_MAX, d2_before = 0.0;
        const double c = 0.4 / resolution;
        if (SORTED) {
            float pi_[3];
            double ni_[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { pi_[d] = xyz[3 * i + d]; ni_[d] = nrm[3 * i + d]; }
            int64_t qn[CH];  // the NEXT chunk's indices
#pragma unroll
            for (int t = 0; t < CH; ++t) qn[t] = t < k ? knnT[(int64_t)t * n + i] : -1;
            for (int j0 = 0; j0 < k; j0 += CH) {
                int64_t q[CH];
                float p[CH][3];
                double nn[CH][3];
#pragma unroll
                for (int t = 0; t < CH; ++t) q[t] = qn[t] < 0 ? i : qn[t];
#pragma unroll
                for (int t = 0; t < CH; ++t)
#pragma unroll
                    for (int d = 0; d < 3; ++d) { p[t][d] = xyz[3 * q[t] + d]; nn[t][d] = nrm[3 * q[t] + d]; }
#pragma unroll
                for (int t = 0; t < CH; ++t) qn[t] = j0 + CH + t < k ? knnT[(int64_t)(j0 + CH + t) * n + i] : -1;
                bool done = false;
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    if (q[t] == i || done) continue;
                    const double t1 = (double)pi_[0] - p[t][0], t2 = (double)pi_[1] - p[t][1], t3 = (double)pi_[2] - p[t][2];
                    const double bound = best * 1.000001 + 1e-15;
                    if (c * c * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound) { done = true; continue; }  // (and so is every neighbour after it)
                    const double m = sv_metric_vals(pi_, ni_, p[t], nn[t], resolution);
                    best = m < best ? m : best;
                }
                if (done) break;
            }
            dis0[i] = best;
            continue;
        }
        const double xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2];
        for (int j0 = 0; j0 < k; j0 += CH) {
            int64_t q[CH];
            float p[CH][3];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                q[t] = j0 + t < k ? knnT[(int64_t)(j0 + t) * n + i] : -1;
                if (q[t] < 0) q[t] = i;  // (a negative entry = "no neighbour here": a point outside the caller's slab)
            }
#pragma unroll
            for (int t = 0; t < CH; ++t) { p[t][0] = xyz[3 * q[t]]; p[t][1] = xyz[3 * q[t] + 1]; p[t][2] = xyz[3 * q[t] + 2]; }
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                if (q[t] == i) continue;
                // sv_metric_at_least(xyz, i, q, resolution, best) on the coordinates already here
                const double t1 = xi - p[t][0], t2 = yi - p[t][1], t3 = zi - p[t][2], bound = best * 1.000001 + 1e-15;
                const double d2 = t1 * t1 + t2 * t2 + t3 * t3;
                unsorted = unsorted || d2 < d2_before * 0.999999;
                d2_before = d2;
                if (c * c * d2 > bound * bound) continue;
                const double m = sv_metric(xyz, nrm, i, q[t], resolution);
                best = m < best ? m : best;
            }
        }
        dis0[i] = best;
    }
    if (!SORTED && __ballot(unsorted) != 0ULL && lane_id() == 0) atomicOr(&st->unsorted, 1);
}
__global__ void start_kernel(State *st, const double *__restrict__ median) {
    const double med = median[0];  // median.h:27-30: nth_element at size / 2 (the element of that rank: select.hip)
    st->lambda0 = med > DBL_EPSILON ? med : DBL_EPSILON;
}
__global__ void init_points_kernel(int64_t n, int32_t *__restrict__ parent, int32_t *__restrict__ size,
                                   unsigned long long *__restrict__ bestm, int32_t *__restrict__ bestu) {
    SV_FOR(i, n) { parent[i] = (int32_t)i; size[i] = 1; bestm[i] = ~0ULL; bestu[i] = NONE; }
}

// What a pass over an edge list needs to know about an end point, in ONE 16-byte record (one load instruction): the pass is bound
// by the scattered lanes it addresses and the cache lines they touch, not by bytes -- representative, the representative's size
// and position through parent[], size[] and xyz[] would be five loads of which three depend on the first.  The position is
// quantised (20 bits per axis of the cloud's bounding cube: 0.6 mm on a 600 m tile): it only serves the test
// "the distance term alone rules the edge out", taken with three steps of slack per axis, i.e. on a LOWER bound of the distance; the
// edges it cannot rule out (a tenth) take the exact test on the float coordinates.  Bits 60..62 of the position word: the coins
// the representative draws in the three sub-rounds of the round the record was made for (heads(), on the caller's index).
struct __attribute__((aligned(16))) Node { int32_t root, size; unsigned int qlo, qhi; };
struct Quant { float mn[3]; float inv_step, step; };
constexpr float QUANT_MAX = 1048575.f;  // 2^20 - 1
__device__ __forceinline__ Quant quant_of(const State *st);
__device__ __forceinline__ void node_pack(Node &nd, const Quant &q, float x, float y, float z, unsigned int coins) {
    const float f[3] = {x, y, z};
    unsigned long long w = (unsigned long long)(coins & 7u) << 60;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float t = (f[d] - q.mn[d]) * q.inv_step;
        t = t < 0.f ? 0.f : (t > QUANT_MAX ? QUANT_MAX : t);
        w |= (unsigned long long)(unsigned int)t << (20 * d);
    }
    nd.qlo = (unsigned int)w;
    nd.qhi = (unsigned int)(w >> 32);
}
__device__ __forceinline__ unsigned int node_coins(const Node &a) { return (a.qhi >> 28) & 7u; }
// the sub-rounds in which an edge u -> v can make an offer: u heads and v tails (bit s: sub-round s)
__device__ __forceinline__ unsigned int edge_eligible(const Node &u, const Node &v) { return node_coins(u) & ~node_coins(v) & 7u; }
// a lower bound of the squared distance between two records' positions, in units of the quantisation step squared
__device__ __forceinline__ float node_dist2_low(const Node &a, const Node &b) {
    const unsigned long long wa = ((unsigned long long)a.qhi << 32) | a.qlo, wb = ((unsigned long long)b.qhi << 32) | b.qlo;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int da = (int)((wa >> (20 * d)) & 0xfffffULL), db = (int)((wb >> (20 * d)) & 0xfffffULL);
        int g = da > db ? da - db : db - da;
        g = g > 3 ? g - 3 : 0;  // (a quantised coordinate lies within a step and a half of the true one, float rounding at 2^20 included: three steps of slack per axis)
        s += (float)g * (float)g;
    }
    return s;
}

__device__ __forceinline__ Quant quant_of(const State *st) {
    Quant q;
    float ext = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        q.mn[d] = ord2f(st->pb[d]);
        const float e = ord2f(st->pb[3 + d]) - q.mn[d];
        ext = e > ext ? e : ext;
    }
    q.step = ext > 0.f ? ext / (QUANT_MAX - 1.f) : 1.f;
    q.inv_step = 1.f / q.step;
    return q;
}

struct SegArgs {
    const float *xyz;
    const double *nrm;
    const int32_t *knnT;        // neighbour lists, transposed
    const int32_t *orig;        // the caller's index of the point at every place (the segmentation works on a spatially sorted copy)
    int64_t n;
    int k;
    double resolution;
    State *st;
    unsigned long long *edges_a, *edges_b, *table, *akey, *bestm, *prop_key;
    double *am, *dis;
    int32_t *parent, *size, *bestu, *prop_u, *prop_v, *la, *lb;
    int32_t *reps_a, *reps_b;  // the lists of node_body (idle arrays: the overflow pass's proposals, the relabelling's ranks)
    Node *node;
    unsigned char *d0, *d1;
};

__device__ __forceinline__ bool fusing(const State *st) { return st->live > st->K && !st->stalled && !st->overflow && !st->overflow_offers; }
__device__ __forceinline__ double lambda_of(const State *st, int r) { return ldexp(st->lambda0, r); }  // :117 `lambda *= 2.0`, exact
// the list of round r (r >= 1) alternates between the two edge buffers
__device__ __forceinline__ unsigned long long *list_of(const SegArgs &a, int r) { return (r & 1) ? a.edges_a : a.edges_b; }
__device__ __forceinline__ unsigned long long list_in_count(const SegArgs &a, int r) {
    return r <= KNN_ROUNDS ? (unsigned long long)a.n * (unsigned long long)a.k : a.st->ne[r - 1];
}
// slots of the filter that merges the parallel edges of a list of n_edges
__host__ __device__ __forceinline__ unsigned long long table_size(unsigned long long n_edges) {
    return n_edges < 4096 ? 1024 : (n_edges / 4 < 0xffffffffULL ? n_edges / 4 : 0xffffffffULL);  // (indexed by the high word of a 32-bit product)
}

// ---- once per lambda round: the round's list and its active edges ------------------------------------------------------
// Every point's record for the round: its representative (the forest is flattened on the way), and that one's size and position.
__device__ __forceinline__ void node_body(const SegArgs &a, int r) {
    int32_t *__restrict__ parent = a.parent;
    // a round that reads a LIST only meets the representatives of the round before as end points: a point that was absorbed
    // earlier (its record of the last pass names another point) is never looked up again.  Those points are kept in a list of
    // their own from the last all-points pass on (round 4 read all n records every round to find them: 160 MB per round and
    // 10 M points): every pass writes the list of the next, the points it finds still their own representative.
    const bool from_list = r > KNN_ROUNDS, to_list = r >= KNN_ROUNDS;
    const int32_t *__restrict__ lin = (r & 1) ? a.reps_a : a.reps_b;
    int32_t *__restrict__ lout = (r & 1) ? a.reps_b : a.reps_a;
    const int64_t count = from_list ? (int64_t)a.st->nrep[r] : a.n;
    const Quant qt = quant_of(a.st);
    __shared__ AppendScratch scratch;
    constexpr int NI = 4;  // points per thread and trip: four independent chains of loads, a quarter of the barriers of the append
    const int64_t step = (int64_t)blockDim.x * NI, stride = (int64_t)gridDim.x * step;
    for (int64_t i0 = (int64_t)blockIdx.x * step; i0 < count; i0 += stride) {  // whole workgroups iterate together
        bool keep[NI], none[NI];
        int32_t i[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int64_t t = i0 + (int64_t)j * blockDim.x + threadIdx.x;
            keep[j] = none[j] = false;
            i[j] = 0;
            if (t < count) {
                i[j] = from_list ? lin[t] : (int32_t)t;
                int32_t q = parent[i[j]];
                while (parent[q] != q) q = parent[q];  // (roots are stable while this pass runs)
                if (q != parent[i[j]]) parent[i[j]] = q;
                Node nd;
                nd.root = q;
                nd.size = a.size[q];
                const int32_t oq = a.orig[q];  // (the coins are drawn on the CALLER's indices)
                const unsigned int coins = (heads(oq, SUBROUNDS * r) ? 1u : 0u) | (heads(oq, SUBROUNDS * r + 1) ? 2u : 0u) | (heads(oq, SUBROUNDS * r + 2) ? 4u : 0u);
                node_pack(nd, qt, a.xyz[3 * (int64_t)q], a.xyz[3 * (int64_t)q + 1], a.xyz[3 * (int64_t)q + 2], coins);
                a.node[i[j]] = nd;
                keep[j] = q == i[j];
            }
        }
        if (to_list) {  // (uniform)
            unsigned long long at[NI], unused[NI];
            block_append2<NI>(&a.st->nrep[r + 1], keep, &a.st->nrep[r + 1], none, at, unused, scratch);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (keep[j]) lout[at[j]] = i[j];
        }
    }
}
__global__ void node_kernel(SegArgs a, int r) {
    if (!fusing(a.st)) return;
    node_body(a, r);
}
// Round r's list from round r - 1's (FROM_KNN: from the neighbour lists themselves, rounds 0 and 1), and its active edges.
// Round 0 has nothing to re-point or merge: only the active edges are picked.
#ifndef SVX_ITEMS
#define SVX_ITEMS 4
#endif
constexpr int BUILD_ITEMS = SVX_ITEMS;   // edges per thread and trip: four independent chains of dependent loads in flight, a quarter of the barriers and counter updates
__device__ __forceinline__ void build_body(const SegArgs &a, int r) {
    State *st = a.st;
    const unsigned long long ne_in = list_in_count(a, r);
    const unsigned long long ts = table_size(ne_in);
    const double lambda = lambda_of(st, r), resolution = a.resolution;
    const unsigned long long *__restrict__ lin = list_of(a, r - 1);
    unsigned long long *__restrict__ lout = list_of(a, r);
    const float *__restrict__ xyz = a.xyz;
    const unsigned int ts32 = (unsigned int)ts;
    const unsigned long long stamp = round_stamp(r);  // (what this round leaves in the device-wide filter is its own)
    // the distance term of the metric alone against lambda, c * d * sizes[v] > lambda, first in float with a margin a thousand
    // times the float error (the few edges inside the margin take the exact test in double)
    const float cfs = (float)(0.4 / resolution) * quant_of(st).step, lambda_hi = (float)(lambda * 1.001);  // (cfs: c times the quantisation step)
    const double cd = 0.4 / resolution;
    const bool emit = r >= KNN_ROUNDS;  // this round writes a list (and passes the device-wide filter)
    __shared__ AppendScratch scratch;
    __shared__ unsigned long long wave_filter[16 * WAVE_FILTER];
    for (int t = (int)threadIdx.x; t < 16 * WAVE_FILTER; t += (int)blockDim.x) wave_filter[t] = DEAD;
    __syncthreads();
    unsigned long long *wf = wave_filter + (threadIdx.x >> 6) * WAVE_FILTER;
    const unsigned long long step = (unsigned long long)blockDim.x * BUILD_ITEMS, stride = (unsigned long long)gridDim.x * step;
    for (unsigned long long e0 = (unsigned long long)blockIdx.x * step; e0 < ne_in; e0 += stride) {  // whole workgroups iterate together
        int32_t u[BUILD_ITEMS], v[BUILD_ITEMS];
        bool live[BUILD_ITEMS], fresh[BUILD_ITEMS], act[BUILD_ITEMS], listed[BUILD_ITEMS];
        unsigned long long key[BUILD_ITEMS];
        double m[BUILD_ITEMS];
        int32_t szv[BUILD_ITEMS];
        unsigned int hsh[BUILD_ITEMS];
        // the edges
#pragma unroll
        for (int j = 0; j < BUILD_ITEMS; ++j) {
            const unsigned long long e = e0 + (unsigned long long)j * blockDim.x + threadIdx.x;
            live[j] = e < ne_in;
            fresh[j] = act[j] = false;
            key[j] = DEAD;
            m[j] = 0.0;
            u[j] = v[j] = 0;
            if (live[j]) {
                const unsigned long long old = lin[e];
                u[j] = key_u(old);
                v[j] = key_v(old);
            }
        }
        // ... re-pointed (the records of the round: one load per end gives representative, size and position)
        int32_t szu[BUILD_ITEMS];
        float d2low[BUILD_ITEMS];
        unsigned int elig[BUILD_ITEMS];  // the sub-rounds in which the edge can make an offer (none: it is never active)
        {
            Node nu[BUILD_ITEMS], nv[BUILD_ITEMS];
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) { nu[j] = a.node[u[j]]; nv[j] = a.node[v[j]]; }
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) {
                u[j] = nu[j].root; v[j] = nv[j].root;
                szu[j] = nu[j].size; szv[j] = nv[j].size;
                d2low[j] = node_dist2_low(nu[j], nv[j]);
                elig[j] = edge_eligible(nu[j], nv[j]);
            }
        }
        // ... merged.  First by the wave's own filter in LDS (direct mapped: an edge that finds itself in its slot is parallel to
        // one this wave has already passed on): the neighbours of a point lie in a few supervoxels, so most of a row's edges are
        // parallel to one another.  Then, for a list: an edge between two representatives without members cannot be any other
        // edge of the list too; the others pass the device-wide filter, one exchange with the slot the edge hashes to.  Both
        // filters forget an edge that another one displaced, so some parallel edges stay: nothing downstream counts edges (minimum
        // metric, smallest index, proposals per absorbed representative), a parallel edge only costs its slot in the next pass.
        // One access per edge and no probe chains instead of a hash set's two or more; the table is a quarter of the list's size
        // and stays in the last-level cache.
        {
            unsigned long long old[BUILD_ITEMS];
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) {
                live[j] = live[j] && u[j] != v[j];
                key[j] = ((unsigned long long)(unsigned int)u[j] << 32) | (unsigned int)v[j];
                unsigned int h = (unsigned int)u[j] * 0x9E3779B1u ^ ((unsigned int)v[j] * 0x85EBCA6Bu + 0x7F4A7C15u);
                h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
                hsh[j] = h;
                if (live[j] && r > 0)
                    live[j] = __hip_atomic_exchange(&wf[h >> 24], key[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != key[j];
                fresh[j] = live[j] && (!emit || (szu[j] == 1 && szv[j] == 1));
                old[j] = DEAD;
                if (live[j] && !fresh[j]) old[j] = atomicExch(&a.table[__umulhi(h * 0x9E3779B1u, ts32)], key[j] | stamp);
            }
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) fresh[j] = fresh[j] || (live[j] && old[j] != (key[j] | stamp));
        }
        // ... and the active ones among the fresh: those the criterion `lambda - sizes[v] * metric(u, v) > 0` (:147-149) accepts
        // with the size v has NOW -- sizes only grow, so no other edge can be accepted in this round.  The distance term of the
        // metric alone may already reach lambda / sizes[v] (with room for every rounding, and for a normal term that rounds a
        // hair below zero): such an edge is not active, and the normals stay unfetched
        {
            bool near[BUILD_ITEMS];
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) {
                const float fs = cfs * (float)szv[j];
                near[j] = fresh[j] && elig[j] != 0u && !(fs * fs * d2low[j] > lambda_hi * lambda_hi);
                // (exactly: the distance term against lambda / sizes[v], without the division)
                if (near[j]) near[j] = !sv_metric_at_least_sized(xyz, u[j], v[j], cd * (double)szv[j], lambda * 1.000001 + (double)szv[j] * 1e-15);
            }
#pragma unroll
            for (int j = 0; j < BUILD_ITEMS; ++j) {
                if (near[j]) {
                    m[j] = sv_metric(xyz, a.nrm, u[j], v[j], resolution);
                    act[j] = lambda - (double)szv[j] * m[j] > 0.0;
                    // (a round without a list: its few active edges pass the device-wide filter here, or the active list of round 2
                    //  would hold every edge four times over -- one per member pair of the two supervoxels)
                    if (act[j] && !emit && r > 0) act[j] = atomicExch(&a.table[__umulhi(hsh[j] * 0x9E3779B1u, ts32)], key[j] | stamp) != (key[j] | stamp);
                }
                listed[j] = fresh[j] && r > 0;  // (counted from round 1 on: an empty list means a disconnected graph; written from round KNN_ROUNDS on)
            }
        }
        unsigned long long at0[BUILD_ITEMS], at1[BUILD_ITEMS];
#ifdef SVX_NO_APPEND  // (timing experiment only: nothing is listed)
        bool any = false;
        for (int j = 0; j < BUILD_ITEMS; ++j) { any = any || listed[j] || act[j]; listed[j] = act[j] = false; at0[j] = at1[j] = 0; }
        if (__ballot(any) == 0x123456789ULL) st->tau_excl = 1;
#else
        block_append2<BUILD_ITEMS>(&st->ne[r], listed, &st->na[r], act, at0, at1, scratch);
#endif
#pragma unroll
        for (int j = 0; j < BUILD_ITEMS; ++j) {
            if (listed[j] && emit) lout[at0[j]] = key[j];
            if (act[j]) { a.akey[at1[j]] = key[j] | ((unsigned long long)elig[j] << 60); a.am[at1[j]] = m[j]; }
        }
    }
}
__global__ __launch_bounds__(1024) void build_kernel(SegArgs a, int r) {
    if (!fusing(a.st)) return;
    build_body(a, r);
}

// ---- rounds 0 .. KNN_ROUNDS: straight from the neighbour lists, one lane per point ------------------------------------------------
// The point's record is one coalesced load, its neighbours come from the transposed lists (coalesced), and the one scattered
// load per edge is the neighbour's record.  A lane notes which of its row's edges are listed / active in two bit masks, the
// workgroup turns the lanes' counts into places in the lists (one counter update per workgroup), and the lanes write.
// Per-lane inclusive scan over the wave + the workgroup's one update of each counter: this lane's first places.
struct CountScratch { int32_t c0[17], c1[17]; unsigned long long base[2]; };
__device__ __forceinline__ void block_reserve2(unsigned long long *counter0, int n0, unsigned long long *counter1, int n1,
                                               unsigned long long &at0, unsigned long long &at1, CountScratch &s) {
    int x0 = n0, x1 = n1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y0 = __shfl_up(x0, d, 64), y1 = __shfl_up(x1, d, 64);
        if (lane_id() >= d) { x0 += y0; x1 += y1; }
    }
    const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 63) { s.c0[wave] = x0; s.c1[wave] = x1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot0 = 0, tot1 = 0;
        for (int w = 0; w < nw; ++w) {
            const int32_t a = s.c0[w], b = s.c1[w];
            s.c0[w] = tot0; s.c1[w] = tot1;
            tot0 += a; tot1 += b;
        }
        s.base[0] = tot0 ? atomicAdd(counter0, (unsigned long long)tot0) : 0ULL;
        s.base[1] = tot1 ? atomicAdd(counter1, (unsigned long long)tot1) : 0ULL;
    }
    __syncthreads();
    at0 = s.base[0] + (unsigned long long)(s.c0[wave] + x0 - n0);
    at1 = s.base[1] + (unsigned long long)(s.c1[wave] + x1 - n1);
    __syncthreads();  // (the scratch is reused by the next call)
}
__device__ __forceinline__ unsigned int edge_hash(int32_t u, int32_t v) {
    unsigned int h = (unsigned int)u * 0x9E3779B1u ^ ((unsigned int)v * 0x85EBCA6Bu + 0x7F4A7C15u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ unsigned int edge_slot(int32_t u, int32_t v, unsigned int ts32) {
    return __umulhi(edge_hash(u, v) * 0x9E3779B1u, ts32);
}
// The workgroup's own filter of the round that writes the first list (r = KNN_ROUNDS), in LDS, in front of the device-wide one:
// by then a supervoxel holds a dozen points, so the 256 consecutive points (Z-curve order) a workgroup takes at a time belong to a
// few dozen supervoxels, and nine of ten of their rows' edges repeat an edge the workgroup has just passed on.  Every one of those
// used to be an exchange with a random word of a table of n k / 4 words (600 MB at 10 M points): a 64-byte read-modify-write at
// the memory side each -- that round alone took 4.7 of the 9.4 ms of the four neighbour-list rounds (profiles/r4_e_sv_trace_10M.log).
// Direct mapped like the others: an edge that finds itself in its slot is parallel to one already listed in this launch and is
// dropped; what another edge displaced is forgotten (it then meets the device-wide filter: nothing is lost but a slot).
constexpr int ROW_FILTER = 2048;  // slots (16 KB of LDS)
constexpr int ROW_SEEN = 8;
#ifndef SVX_ROW_CHUNK
#define SVX_ROW_CHUNK 10
#endif
constexpr int ROW_CHUNK = SVX_ROW_CHUNK;
#ifndef SVX_NEAR_BATCH
#define SVX_NEAR_BATCH 4
#endif
constexpr int NEAR_BATCH = SVX_NEAR_BATCH;  // near edges of a row that are measured together
constexpr int LIST_BATCH = 5;  // listed edges of a row whose representatives are fetched together
#ifndef SVX_SWEEP_CHUNK
#define SVX_SWEEP_CHUNK 10
#endif
constexpr int SWEEP_CHUNK = SVX_SWEEP_CHUNK;  // neighbours of a point whose labels are fetched together
constexpr int FOREIGN = 3;       // foreign labels of a point that are measured together
__device__ __forceinline__ void rows_body(const SegArgs &a, int r) {
    State *st = a.st;
    const int64_t n = a.n;
    const int k = a.k;
    const bool emit = r >= KNN_ROUNDS;  // this round writes a list (and passes the device-wide filter)
    const double lambda = lambda_of(st, r), lambda0 = st->lambda0, resolution = a.resolution;
    const unsigned int ts32 = (unsigned int)table_size((unsigned long long)n * (unsigned long long)k);
    const unsigned long long stamp = round_stamp(r);  // (what this round leaves in the device-wide filter is its own)
    // the distance term of the metric alone against lambda, c * d * sizes[v] > lambda, first in float with a margin a thousand
    // times the float error (the few edges inside the margin take the exact test in double)
    const float cfs = (float)(0.4 / resolution) * quant_of(st).step, lambda_hi = (float)(lambda * 1.001);  // (cfs: c times the quantisation step)
    const double cd = 0.4 / resolution;
    // Round 0: every point is its own representative and its list is in ascending order of distance (what a neighbour search
    // returns; min_metric_kernel checked it), so the first neighbour the distance term rules out ends the row -- at lambda0, the
    // median of the smallest metrics, that is among the first few, and whole waves leave after the first chunk.
    const bool stop_early = r == 0 && !st->unsorted;
    unsigned long long *__restrict__ lout = list_of(a, r);
    const float *__restrict__ xyz = a.xyz;
    const int32_t *__restrict__ knnT = a.knnT;
    __shared__ CountScratch scratch;
    __shared__ unsigned long long row_filter[ROW_FILTER];
    if (emit) {
        for (int t = (int)threadIdx.x; t < ROW_FILTER; t += (int)blockDim.x) row_filter[t] = DEAD;
        __syncthreads();
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {  // whole workgroups iterate together
        const int64_t i = i0 + threadIdx.x;
        unsigned long long listmask = 0ULL, actmask = 0ULL;
        int n_fresh = 0;
        int32_t u = 0;
        int kept_j[NEAR_BATCH];  // the first batch of the row's near edges that turned out active: slot, key (with its sub-round bits), metric
        unsigned long long kept_key[NEAR_BATCH];
        double kept_m[NEAR_BATCH];
#pragma unroll
        for (int t = 0; t < NEAR_BATCH; ++t) { kept_j[t] = -1; kept_key[t] = 0ULL; kept_m[t] = 0.0; }
        // (round 0: a point whose smallest metric is not below lambda0 has no active edge -- half of the points)
        const bool row = i < n && !(r == 0 && !(a.dis[i] < lambda0));
        if (row) {
            const Node nu = a.node[i];
            u = nu.root;
            // pass 1, every neighbour, cheap: the edges that are new to this row, pass the filter, and that the quantised
            // positions cannot rule out ("near").  Pass 2 takes the exact test and the metric for the near ones only: in one
            // loop every step would pay for them -- some lane of the wave always has a near edge -- although a row has one or two.
            unsigned long long nearmask = 0ULL;
            int32_t seen[ROW_SEEN];
#pragma unroll
            for (int t = 0; t < ROW_SEEN; ++t) seen[t] = u;
            int at_seen = 0;
            // (the neighbours ROW_CHUNK at a time: their indices, then their records, are loaded together -- one after the other
            //  a row would wait for sixty memory round trips in turn, and the pass is bound by exactly that wait)
            int32_t qn[ROW_CHUNK];  // the NEXT chunk's indices: requested together with the current chunk's records
#pragma unroll
            for (int c = 0; c < ROW_CHUNK; ++c) qn[c] = c < k ? knnT[(int64_t)c * n + i] : -1;
            for (int j0 = 0; j0 < k; j0 += ROW_CHUNK) {
                int32_t q[ROW_CHUNK];
                Node nq[ROW_CHUNK];
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) {
                    q[c] = qn[c];
                    if (q[c] < 0 || (int64_t)q[c] >= n) q[c] = (int32_t)i;  // (its own record: the edge is dropped as a self loop)
                }
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) nq[c] = a.node[q[c]];
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) qn[c] = j0 + ROW_CHUNK + c < k ? knnT[(int64_t)(j0 + ROW_CHUNK + c) * n + i] : -1;
                // new to this row?  (its own supervoxel, or one this row has just met, is not)
                bool fresh[ROW_CHUNK], asked[ROW_CHUNK];
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) {
                    const int32_t v = nq[c].root;
                    bool dup = q[c] == (int32_t)i;
#pragma unroll
                    for (int t = 0; t < ROW_SEEN; ++t) dup = dup || seen[t] == v;
                    if (!dup) {
#pragma unroll
                        for (int t = 0; t < ROW_SEEN; ++t) seen[t] = t == at_seen ? v : seen[t];  // (a ring: the oldest entry goes)
                        at_seen = at_seen + 1 == ROW_SEEN ? 0 : at_seen + 1;
                    }
                    fresh[c] = !dup;
                    // a list: an edge between two representatives without members cannot be any other edge of the list too; the
                    // others pass the device-wide filter (see build_body) -- the chunk's exchanges are issued together
                    asked[c] = fresh[c] && emit && !(nu.size == 1 && nq[c].size == 1);
                    if (asked[c]) {  // the workgroup's own filter first (see ROW_FILTER)
                        const unsigned long long key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)v;
                        if (__hip_atomic_exchange(&row_filter[edge_hash(u, v) & (ROW_FILTER - 1)], key, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP) == key) {
                            asked[c] = false;
                            fresh[c] = false;
                        }
                    }
                }
                unsigned long long found[ROW_CHUNK];
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) {
                    const unsigned long long key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)nq[c].root;
                    found[c] = asked[c] ? atomicExch(&a.table[edge_slot(u, nq[c].root, ts32)], key | stamp) : DEAD;
                }
                bool far_last = false;
#pragma unroll
                for (int c = 0; c < ROW_CHUNK; ++c) {
                    const int j = j0 + c;
                    const unsigned long long key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)nq[c].root;
                    if (fresh[c] && found[c] != (key | stamp)) {
                        ++n_fresh;
                        if (emit) listmask |= 1ULL << j;
                        const float fs = cfs * (float)nq[c].size;
                        const bool far = fs * fs * node_dist2_low(nu, nq[c]) > lambda_hi * lambda_hi;
                        // (an edge that can make no offer in any of the round's sub-rounds -- u tails or v heads in all three -- is never active)
                        if (!far && edge_eligible(nu, nq[c]) != 0u) nearmask |= 1ULL << j;
                        far_last = far;  // (of the last REAL neighbour of the chunk: padding and "no neighbour here" are not fresh)
                    }
                }
                if (stop_early && far_last) break;
            }
            // The near edges (a row has one or two): exact distance test, metric, criterion -- NEAR_BATCH at a time, so that the
            // loads of a batch (neighbour index -> its record -> its representative's position and normal) are three round trips
            // for the batch, not four for every edge in turn (round 4: this loop and the recomputation at the writes below were
            // most of the rounds' time; the row's own position and normal are the same for all of its edges).
            if (nearmask) {
                float pu[3];
                double nu_[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { pu[d] = xyz[3 * (int64_t)u + d]; nu_[d] = a.nrm[3 * (int64_t)u + d]; }
                bool first = true;
                while (nearmask) {
                    int js[NEAR_BATCH];
                    bool on[NEAR_BATCH];
                    int32_t qv[NEAR_BATCH];
                    Node nv[NEAR_BATCH];
                    float pv[NEAR_BATCH][3];
                    double nn[NEAR_BATCH][3];
#pragma unroll
                    for (int t = 0; t < NEAR_BATCH; ++t) {
                        on[t] = nearmask != 0ULL;
                        js[t] = on[t] ? __ffsll((long long)nearmask) - 1 : 0;
                        nearmask &= nearmask - 1ULL;  // (0 stays 0)
                    }
#pragma unroll
                    for (int t = 0; t < NEAR_BATCH; ++t) qv[t] = on[t] ? knnT[(int64_t)js[t] * n + i] : (int32_t)i;
#pragma unroll
                    for (int t = 0; t < NEAR_BATCH; ++t) nv[t] = a.node[qv[t]];
#pragma unroll
                    for (int t = 0; t < NEAR_BATCH; ++t)
#pragma unroll
                        for (int d = 0; d < 3; ++d) { pv[t][d] = xyz[3 * (int64_t)nv[t].root + d]; nn[t][d] = a.nrm[3 * (int64_t)nv[t].root + d]; }
#pragma unroll
                    for (int t = 0; t < NEAR_BATCH; ++t) {
                        if (!on[t]) continue;
                        const int32_t v = nv[t].root;
                        // sv_metric_at_least_sized(xyz, u, v, cd * size, lambda * 1.000001 + size * 1e-15) on the values already here
                        const double t1 = (double)pu[0] - pv[t][0], t2 = (double)pu[1] - pv[t][1], t3 = (double)pu[2] - pv[t][2];
                        const double cs = cd * (double)nv[t].size, bound = lambda * 1.000001 + (double)nv[t].size * 1e-15;
                        if (cs * cs * (t1 * t1 + t2 * t2 + t3 * t3) > bound * bound) continue;
                        const double m = sv_metric_vals(pu, nu_, pv[t], nn[t], resolution);
                        if (!(lambda - (double)nv[t].size * m > 0.0)) continue;
                        // (a round without a list: its few active edges pass the device-wide filter here)
                        const unsigned long long key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)v;
                        if (!emit && r > 0 && atomicExch(&a.table[edge_slot(u, v, ts32)], key | stamp) == (key | stamp)) continue;
                        actmask |= 1ULL << js[t];
                        if (first) {  // (the first batch's results are kept for the writes below)
                            kept_j[t] = js[t];
                            kept_key[t] = key | ((unsigned long long)edge_eligible(nu, nv[t]) << 60);
                            kept_m[t] = m;
                        }
                    }
                    first = false;
                }
            }
        }
        // (the list's length is counted from round 1 on -- an empty list means a disconnected graph -- and written from round KNN_ROUNDS on)
        unsigned long long at0, at1;
        block_reserve2(&st->ne[r], r > 0 ? n_fresh : 0, &st->na[r], (int)__popcll(actmask), at0, at1, scratch);
        while (listmask) {  // (LIST_BATCH edges at a time: index, then record, are two round trips per batch)
            int32_t ql[LIST_BATCH], vl[LIST_BATCH];
            bool on[LIST_BATCH];
#pragma unroll
            for (int t = 0; t < LIST_BATCH; ++t) {
                on[t] = listmask != 0ULL;
                const int j = on[t] ? __ffsll((long long)listmask) - 1 : 0;
                listmask &= listmask - 1ULL;
                ql[t] = on[t] ? knnT[(int64_t)j * n + i] : (int32_t)i;
            }
#pragma unroll
            for (int t = 0; t < LIST_BATCH; ++t) vl[t] = a.node[ql[t]].root;
#pragma unroll
            for (int t = 0; t < LIST_BATCH; ++t)
                if (on[t]) lout[at0++] = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)vl[t];
        }
        while (actmask) {
            const int j = __ffsll((long long)actmask) - 1;
            actmask &= actmask - 1ULL;
            unsigned long long key = 0ULL;
            double m = 0.0;
            bool have = false;
#pragma unroll
            for (int t = 0; t < NEAR_BATCH; ++t)
                if (kept_j[t] == j) { key = kept_key[t]; m = kept_m[t]; have = true; }
            if (!have) {  // (a row with more active edges than one batch holds: measured again)
                const Node nv = a.node[knnT[(int64_t)j * n + i]];
                const int32_t v = nv.root;
                key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)v | ((unsigned long long)edge_eligible(a.node[i], nv) << 60);
                m = sv_metric(xyz, a.nrm, u, v, resolution);
            }
            a.akey[at1] = key;
            a.am[at1++] = m;
        }
    }
}
#ifndef SVX_ROWS_WPE
#define SVX_ROWS_WPE 1
#endif
__global__ __launch_bounds__(256, SVX_ROWS_WPE) void rows_kernel(SegArgs a, int r) {
    if (!fusing(a.st)) return;
    rows_body(a, r);
}

// ---- one sub-round (number rho = SUBROUNDS * r + s) of conflict-free fusion over the active edges of round r ----------------
// cand:   an active edge u -> v whose ends are both still representatives, with u heads, v tails and
//         sizes[v] * metric(u, v) < lambda, offers u to v: bestm[v] = min metric (atomicMin on the ordered image of the double);
//         the offers are listed;
// cand2:  among the offers of smallest metric the smallest u wins: bestu[v] = min u; the first to reach a v counts a proposal;
// apply:  Link(v, bestu[v]) for every proposal -- unless there are more proposals than representatives to spare: then nothing
//         is applied here, every later pass of the fusion returns at once, and the overflow passes after the schedule apply
//         the best (the sub-round's bestm / bestu stay as they are until then).
__device__ __forceinline__ bool offer_of(const SegArgs &a, unsigned long long e, int s, double lambda, int32_t &u, int32_t &v, double &m) {
    const unsigned long long key = a.akey[e];
    // u heads and v tails in sub-round s: bit 60 + s of the key, set when the edge was picked (the coins of the round's
    // representatives travel in their records); three quarters of the edges end here, with one coalesced load
    if (!((key >> (60 + s)) & 1ULL)) return false;
    u = key_u(key);
    v = key_v(key);
    const int32_t pu = a.parent[u], pv = a.parent[v], sz = a.size[v];  // (loaded together, not one behind the other)
    m = a.am[e];
    return pu == u && pv == v && lambda - (double)sz * m > 0.0;  // :147-149 `improvement > 0.0`
}
__device__ __forceinline__ void cand_body(const SegArgs &a, int r, int s) {
    State *st = a.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->live_snap = st->live;  // (the previous sub-round's apply has finished)
        if (s == 0) {
            if (r >= 1 && st->ne[r] == 0ULL) st->stalled = 1;  // disconnected graph of representatives: the reference would never return
            else st->rounds = r + 1;
        }
    }
    const int rho = SUBROUNDS * r + s;
    const double lambda = lambda_of(st, r);
    const unsigned long long na = st->na[r];
    // the offers (their places in the active list) go to the edge buffer that is idle until the next round's list is built:
    // the two passes after this one walk the sub-round's offers, a twentieth of the active edges, not the active edges again
    unsigned long long *__restrict__ offers = list_of(a, r + 1);
    const unsigned long long cap = (unsigned long long)a.n * (unsigned long long)a.k;
    __shared__ AppendScratch scratch;
    constexpr int ITEMS = 4;
    const unsigned long long step = (unsigned long long)blockDim.x * ITEMS, stride = (unsigned long long)gridDim.x * step;
    for (unsigned long long e0 = (unsigned long long)blockIdx.x * step; e0 < na; e0 += stride) {  // whole workgroups iterate together
        bool off[ITEMS], none[ITEMS];
        unsigned long long e[ITEMS], at[ITEMS], unused[ITEMS], mo[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            e[j] = e0 + (unsigned long long)j * blockDim.x + threadIdx.x;
            int32_t u = 0, v = 0;
            double m = 0.0;
            off[j] = e[j] < na && offer_of(a, e[j], s, lambda, u, v, m);
            none[j] = false;
            mo[j] = d2ord(m);
            if (off[j]) atomicMin(&a.bestm[v], mo[j]);
        }
        block_append2<ITEMS>(&st->n_off[rho], off, &st->n_off[rho], none, at, unused, scratch);
        // (an offer is listed with its edge and its metric's ordered image: the two passes below read the list front to back
        //  and need no look-up of the edge)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (off[j]) {
                // (two words per offer in a buffer of n k words: offers are a few per cent of n k on any real cloud; a cloud that
                //  offered more than half of its neighbour edges at once is refused -- status bit 3 -- not written past the end)
                if (2 * at[j] + 1 < cap) { offers[2 * at[j]] = a.akey[e[j]]; offers[2 * at[j] + 1] = mo[j]; }
                else st->overflow_offers = 1;
            }
    }
}
__device__ __forceinline__ void cand2_body(const SegArgs &a, int r, int s) {
    State *st = a.st;
    const int rho = SUBROUNDS * r + s;
    const unsigned long long *__restrict__ offers = list_of(a, r + 1);
    const unsigned long long n_off = st->n_off[rho];
    int first = 0;
    SV_FOR(i, n_off) {
        const unsigned long long key = offers[2 * i], mo = offers[2 * i + 1];
        const int32_t u = key_u(key), v = key_v(key);
        if (a.bestm[v] == mo) first += atomicMin(&a.bestu[v], a.orig[u]) == NONE ? 1 : 0;  // (ties: the smallest index of the caller's)
    }
    block_add(&st->n_prop[rho], first, false);
}
__device__ __forceinline__ void apply_body(const SegArgs &a, int r, int s) {
    State *st = a.st;
    const int rho = SUBROUNDS * r + s;
    if (st->n_prop[rho] > st->live_snap - st->K) {  // (both final since the kernels before: every thread takes the same branch)
        if (blockIdx.x == 0 && threadIdx.x == 0) st->overflow = rho + 1;
        return;
    }
    const unsigned long long *__restrict__ offers = list_of(a, r + 1);
    const unsigned long long n_off = st->n_off[rho];
    int dropped = 0;
    SV_FOR(i, n_off) {
        const unsigned long long key = offers[2 * i], mo = offers[2 * i + 1];
        const int32_t u = key_u(key), v = key_v(key);
        const int32_t ou = a.orig[u];
        if (a.bestu[v] == ou && a.bestm[v] == mo &&
            atomicCAS(&a.bestu[v], ou, NONE) == ou) {  // (the claim lets one of several equal edges through, and leaves bestu clean for the next sub-round)
            a.bestm[v] = ~0ULL;
            a.parent[v] = u;                   // Link(v, u), disjoint_set.h:77-85
            atomicAdd(&a.size[u], a.size[v]);  // sizes[i] += sizes[j], :153
            ++dropped;
        }
    }
    block_add(&st->live, dropped, true);
}
__global__ void cand_kernel(SegArgs a, int r, int s) {
    if (!fusing(a.st)) return;
    cand_body(a, r, s);
}
__global__ void cand2_kernel(SegArgs a, int r, int s) {
    if (!fusing(a.st)) return;
    cand2_body(a, r, s);
}
__global__ void apply_kernel(SegArgs a, int r, int s) {
    if (!fusing(a.st)) return;
    apply_body(a, r, s);
}

// ---- the sub-round that would go below K (at most one per segmentation; after the schedule) ---------------------------------
// collect: its proposals (key = float image of the loss : index); select (one workgroup): exactly the (live - K) smallest keys --
// radix select, 8 passes of 8 bits over the list; apply: those.
__global__ __launch_bounds__(1024) void overflow_collect_kernel(SegArgs a) {
    State *st = a.st;
    if (!st->overflow) return;
    const int rho = st->overflow - 1, r = rho / SUBROUNDS;
    const unsigned long long *__restrict__ offers = list_of(a, r + 1);
    const unsigned long long n_off = st->n_off[rho];
    __shared__ AppendScratch scratch;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x; i0 < n_off; i0 += stride) {  // whole workgroups iterate together
        const unsigned long long i = i0 + threadIdx.x;
        int32_t u = 0, v = 0;
        double m = 0.0;
        bool has = false;
        if (i < n_off) {
            const unsigned long long key = offers[2 * i], mo = offers[2 * i + 1];
            u = key_u(key);
            v = key_v(key);
            m = ord2d(mo);
            // (the list may hold an edge twice: the claim lets one of them through)
            const int32_t ou = a.orig[u];
            has = a.bestu[v] == ou && a.bestm[v] == mo && atomicCAS(&a.bestu[v], ou, NONE) == ou;
        }
        const bool t0[1] = {has}, t1[1] = {false};
        unsigned long long at_[1], unused[1];
        block_append2<1>(&st->n_list, t0, &st->n_list, t1, at_, unused, scratch);
        const unsigned long long at = at_[0];
        if (has) {  // (one winning edge per proposing v: at most n entries)
            const double loss = (double)a.size[v] * m;
            a.prop_key[at] = ((unsigned long long)f2ord((float)loss) << 32) | (unsigned int)a.orig[v];  // (ties by the caller's index)
            a.prop_u[at] = u;
            a.prop_v[at] = v;
        }
    }
}
__global__ __launch_bounds__(1024) void overflow_select_kernel(State *st, const unsigned long long *__restrict__ prop_key) {  // (1024 threads)
    if (!st->overflow) return;
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_rank;
    const int a = (int)st->n_list, budget = st->live - st->K;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { s_prefix = 0ULL; s_rank = budget; }  // 1-based rank of the last key to keep (budget < a here)
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
    if (tid == 0) st->tau_excl = s_prefix + 1ULL;  // keys are unique (index in the low word)
}
__global__ void overflow_apply_kernel(State *st, const unsigned long long *__restrict__ prop_key, const int32_t *__restrict__ prop_u,
                                      const int32_t *__restrict__ prop_v, int32_t *__restrict__ parent, int32_t *__restrict__ size) {
    if (!st->overflow) return;
    const int a = (int)st->n_list;
    const unsigned long long tau = st->tau_excl;
    int dropped = 0;
    SV_FOR(i, a) {
        const unsigned long long key = prop_key[i];
        if (key >= tau) continue;
        const int32_t v = prop_v[i], u = prop_u[i];
        parent[v] = u;
        atomicAdd(&size[u], size[v]);
        ++dropped;
    }
    dropped = wave_sum(dropped);
    if (lane_id() == 0 && dropped) atomicSub(&st->live, dropped);
}

// ---- labels, boundary exchange, relabel ------------------------------------------------------------------------------
__global__ void flatten_kernel(int64_t n, int32_t *__restrict__ parent) {
    SV_FOR(i, n) {
        int32_t r = parent[i];
        while (parent[r] != r) r = parent[r];  // (roots are stable while this pass runs)
        if (r != parent[i]) parent[i] = r;
    }
}
#pragma clang fp contract(off)
__global__ void labels_init_kernel(const float *__restrict__ xyz, const double *__restrict__ nrm, double resolution, int64_t n,
                                   const int32_t *__restrict__ parent, int32_t *__restrict__ la, int32_t *__restrict__ lb,
                                   double *__restrict__ dis) {
    SV_FOR(i, n) {
        const int32_t r = parent[i];
        la[i] = r;
        lb[i] = r;
        dis[i] = sv_metric(xyz, nrm, i, (int64_t)r, resolution);  // :186-189
    }
}
// What sweep number s does, from what sweep s - 1 did (s - 1's slots are final when s starts): a sweep that changed labels is
// followed by one over the dirty points; a sweep over the dirty points that changed nothing by one over every point; a sweep over
// every point that changed nothing ends the relaxation.
__device__ __forceinline__ void sweep_state(const State *st, int s, bool &on, bool &full) {
    if (s == 0) { on = true; full = true; return; }
    const bool pon = st->sw_on[s - 1] != 0, pfull = st->sw_full[s - 1] != 0, pch = st->sw_changed[s - 1] != 0;
    if (pch) { on = pon; full = false; }
    else if (!pfull) { on = pon; full = true; }
    else { on = false; full = true; }
}
// One sweep of the exchange (:214-226 for every point at once, reading the labels of the previous sweep).  A point is
// looked at when it, or a point that lists it or that it lists, changed in the previous sweep (`dirty`); a full sweep
// looks at every point (the first sweep, and the verification sweep that ends the relaxation).
__device__ __forceinline__ void sweep_body(const SegArgs &a, int s) {
    State *st = a.st;
    bool on, full;
    sweep_state(st, s, on, full);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->sw_on[s] = on ? 1 : 0;
        st->sw_full[s] = full ? 1 : 0;
        if (on) st->sweeps_done = s + 1;
    }
    if (!on) return;
    const float *__restrict__ xyz = a.xyz;
    const double *__restrict__ nrm = a.nrm;
    const int32_t *__restrict__ knnT = a.knnT;
    const int64_t n = a.n;
    const int k = a.k;
    const double resolution = a.resolution;
    const bool odd = (s & 1) != 0;
    const int32_t *__restrict__ lin = odd ? a.lb : a.la;
    int32_t *__restrict__ lout = odd ? a.la : a.lb;
    // (a point reads only its OWN flag of the previous sweep and clears it on the way: the buffer is clean again when the
    // next sweep writes its flags into it)
    unsigned char *__restrict__ din = odd ? a.d1 : a.d0;
    unsigned char *__restrict__ dout = odd ? a.d0 : a.d1;
    double *__restrict__ dis = a.dis;
    bool any = false;
    SV_FOR(i, n) {
        const int32_t own = lin[i];
        int32_t bl = own;
        const bool look = full || din[i] != 0;
        din[i] = 0;
        if (look) {
            double best = dis[i];
            // The foreign labels of the row, first in list order, FOREIGN at a time: their representatives' positions and normals
            // are requested together and measured together, then folded in list order with the reference's strict `<` (:219-223:
            // the first of several equal minima wins, the point's own label wins a tie).  A label met again is either the best so
            // far or has lost for good (the best only decreases).  Round 4 measured them one behind the other, two dependent
            // round trips each, with a pre-test on the distance term in between.
            int32_t fb[FOREIGN];
            int nf = 0;
#pragma unroll
            for (int t = 0; t < FOREIGN; ++t) fb[t] = own;
            float pi_[3];
            double ni_[3];
            bool mine = false;  // (the point's own position and normal: fetched when the first foreign label shows up)
            auto flush = [&]() {
                if (!mine) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { pi_[d] = xyz[3 * i + d]; ni_[d] = nrm[3 * i + d]; }
                    mine = true;
                }
                float pb[FOREIGN][3];
                double nb[FOREIGN][3];
#pragma unroll
                for (int t = 0; t < FOREIGN; ++t)
#pragma unroll
                    for (int d = 0; d < 3; ++d) { pb[t][d] = xyz[3 * (int64_t)fb[t] + d]; nb[t][d] = nrm[3 * (int64_t)fb[t] + d]; }
#pragma unroll
                for (int t = 0; t < FOREIGN; ++t) {
                    if (t >= nf) continue;
                    const double d = sv_metric_vals(pi_, ni_, pb[t], nb[t], resolution);
                    if (d < best) { best = d; bl = fb[t]; }
                }
                nf = 0;
            };
            int32_t qn[SWEEP_CHUNK];  // the NEXT chunk's indices: requested together with the current chunk's labels
#pragma unroll
            for (int c = 0; c < SWEEP_CHUNK; ++c) qn[c] = c < k ? knnT[(int64_t)c * n + i] : -1;
            for (int j0 = 0; j0 < k; j0 += SWEEP_CHUNK) {
                int32_t lab[SWEEP_CHUNK];
#pragma unroll
                for (int c = 0; c < SWEEP_CHUNK; ++c) lab[c] = lin[qn[c] >= 0 ? qn[c] : (int32_t)i];
#pragma unroll
                for (int c = 0; c < SWEEP_CHUNK; ++c) qn[c] = j0 + SWEEP_CHUNK + c < k ? knnT[(int64_t)(j0 + SWEEP_CHUNK + c) * n + i] : -1;
                bool again;
                unsigned int settled = 0u;  // places of the chunk that are dealt with (every walk settles at least one: the walks end)
                do {  // (a chunk that meets more new labels than the buffer has room for is walked again after a flush)
                    again = false;
#pragma unroll
                    for (int c = 0; c < SWEEP_CHUNK; ++c) {
                        if (settled & (1u << c)) continue;
                        const int32_t b = lab[c];
                        bool known = b == own || b == bl;
#pragma unroll
                        for (int t = 0; t < FOREIGN; ++t) known = known || fb[t] == b;
                        if (!known && nf == FOREIGN) { again = true; continue; }  // (full: this label and the ones after it wait for the next walk)
                        settled |= 1u << c;
                        if (known) continue;
#pragma unroll
                        for (int t = 0; t < FOREIGN; ++t) fb[t] = t == nf ? b : fb[t];
                        ++nf;
                    }
                    if (again) flush();
                } while (again);
            }
            if (nf > 0) flush();
            if (bl != own) {
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
    if (__ballot(any) != 0ULL && lane_id() == 0) atomicOr(&st->sw_changed[s], 1);
}
__global__ void sweep_kernel(SegArgs a, int s) { sweep_body(a, s); }
__global__ void root_flag_kernel(int64_t n, const int32_t *__restrict__ parent, const int32_t *__restrict__ orig, int32_t *__restrict__ flag) {
    SV_FOR(i, n) flag[orig[i]] = parent[i] == (int32_t)i ? 1 : 0;  // (flag and rank live in the caller's index space)
}
__global__ void relabel_kernel(State *st, int64_t n, const int32_t *__restrict__ l0, const int32_t *__restrict__ l1, const int32_t *__restrict__ orig,
                               const int32_t *__restrict__ flag, const int32_t *__restrict__ rank, int32_t *__restrict__ labels_out,
                               int32_t *__restrict__ reps_out, int32_t *__restrict__ info_out) {
    const int32_t *__restrict__ lab = (st->sweeps_done & 1) ? l1 : l0;
    SV_FOR(i, n) {
        const int32_t oi = orig[i];
        labels_out[oi] = rank[orig[lab[i]]];  // :241-247: position of the representative in ascending index order
        if (reps_out && flag[oi]) reps_out[rank[oi]] = oi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && info_out) {
        bool on, full;
        sweep_state(st, st->sweeps_done, on, full);  // would another sweep run? (only the budget can have stopped it)
        info_out[0] = rank[n - 1] + flag[n - 1];  // supervoxels produced
        info_out[1] = st->K;                      // occupied grid cells (the target)
        info_out[2] = (st->stalled ? 1 : 0) | (st->live > st->K && !st->stalled ? 2 : 0) | (on ? 4 : 0) | (st->overflow_offers ? 8 : 0);
        info_out[3] = st->sweeps_done;
        const unsigned long long lb = (unsigned long long)__double_as_longlong(st->lambda0);
        info_out[4] = (int32_t)(unsigned int)(lb & 0xffffffffULL);  // the fusion's starting lambda (:105-113), the double's two words
        info_out[5] = (int32_t)(unsigned int)(lb >> 32);
        info_out[6] = st->rounds;                 // lambda rounds entered
        info_out[7] = st->overflow;               // 1 + the sub-round that was cut to reach K exactly (0: none was)
    }
}

// ---- whatever the schedule of launches did not cover, in ONE workgroup ------------------------------------------------------
// The host cannot know how many lambda rounds and sweeps a cloud needs (no synchronisation), and a schedule long enough for
// every cloud (56 rounds, 1024 sweeps) is mostly launches that return at once.  So the schedule covers what clouds normally need
// (SCHED_ROUNDS, SCHED_SWEEPS) and the rest -- normally nothing: the kernel returns -- runs here: the same device functions in
// loops that END when the state says so, separated by workgroup barriers.  ONE workgroup: no assumption about co-residency, no
// grid barrier that could wait for a workgroup the device never scheduled (round 2's persistent grid could), at the price of
// one CU's speed for rounds that late -- by then the lists are short (they roughly halve per round) --, and for sweeps beyond
// the scheduled ones (mostly a scan of the dirty flags).
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// fusing(st) as ONE value for the whole workgroup: thread 0 reads the state (written before the last wg_sync) and hands its
// verdict round through LDS.  Every wave reading the state for itself is not enough: the bodies that follow a check write the
// very words it reads (cand_body: `stalled`, `live_snap`), so a wave that arrives late could see another verdict than the waves
// already inside the body and leave the loop while they wait at the next barrier.
__device__ __forceinline__ bool wg_fusing(const State *st) {
    __shared__ int verdict;
    if (threadIdx.x == 0) verdict = fusing(st) ? 1 : 0;
    __syncthreads();
    const bool f = verdict != 0;
    __syncthreads();  // (the next call rewrites the word)
    return f;
}
__global__ __launch_bounds__(1024) void segment_rest_kernel(SegArgs a, int first_round, int first_sweep) {
    State *st = a.st;
    if (first_round >= 0) {
        for (int r = first_round; r < LAMBDA_ROUNDS; ++r) {
            if (!wg_fusing(st)) break;
            node_body(a, r);
            wg_sync();
            if (r <= KNN_ROUNDS) rows_body(a, r);
            else build_body(a, r);
            wg_sync();
            for (int s = 0; s < SUBROUNDS; ++s) {
                if (!wg_fusing(st)) break;
                cand_body(a, r, s);
                wg_sync();
                if (!wg_fusing(st)) break;  // (cand may have found the list empty)
                cand2_body(a, r, s);
                wg_sync();
                apply_body(a, r, s);
                wg_sync();
            }
        }
    } else {
        for (int s = first_sweep; s < SWEEPS; ++s) {
            if (s > 0 && !st->sw_on[s - 1]) break;
            sweep_body(a, s);
            wg_sync();
        }
    }
}

static inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }

struct Ws {
    State *st;
    unsigned long long *keys_a, *keys_b, *edges_a, *edges_b, *table, *akey, *bestm, *prop_key;
    double *am, *dis, *median;
    void *sel;  // select_ranks_f64's workspace
    int32_t *parent, *size, *bestu, *prop_u, *la, *lb, *flag, *rank, *knnT, *orig, *pos_of, *ids_in;
    unsigned int *okey_a, *okey_b;
    float *xyz_p;
    double *nrm_p;
    Node *node;
    unsigned char *d0, *d1;
    void *prim;
    size_t prim_bytes, total;
};
static int layout(int64_t n, int k, Ws &w, unsigned char *base) {
    size_t sort_u = 0, scan_b = 0;
    unsigned long long *u0 = nullptr;
    int32_t *i0 = nullptr;
    if (rocprim::radix_sort_keys(nullptr, sort_u, u0, u0, (size_t)n, 0, 64, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::exclusive_scan(nullptr, scan_b, i0, i0, 0, (size_t)n, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    size_t sort_p = 0;
    unsigned int *k0 = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, sort_p, k0, k0, i0, i0, (size_t)n, 0, 32, 0, false) != hipSuccess) return F4L_EHIP;
    size_t prim = sort_u;
    prim = prim > scan_b ? prim : scan_b;
    prim = prim > sort_p ? prim : sort_p;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    const size_t ne = (size_t)n * (size_t)k;
    w.st = (State *)carve(sizeof(State));
    w.edges_a = (unsigned long long *)carve(ne * 8);
    w.edges_b = (unsigned long long *)carve(ne * 8);
    w.table = (unsigned long long *)carve((ne / 4 > 2 * (size_t)n + 1024 ? ne / 4 : 2 * (size_t)n + 1024) * 8);  // (the filter; the grid keys and sorted metrics: 2 n words)
    // the grid keys and the sorted metrics are dead before the edge table is first used: they alias it
    w.keys_a = w.table;
    w.keys_b = w.table ? w.table + n : nullptr;
    w.akey = (unsigned long long *)carve(ne * 8);
    w.am = (double *)carve(ne * 8);
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
    w.node = (Node *)carve((size_t)n * sizeof(Node));
    w.orig = (int32_t *)carve((size_t)n * 4);
    w.pos_of = (int32_t *)carve((size_t)n * 4);
    w.xyz_p = (float *)carve((size_t)n * 12);
    w.nrm_p = (double *)carve((size_t)n * 24);
    // (sort keys and unsorted ids of the ordering: dead before the label buffers are first used)
    w.okey_a = (unsigned int *)w.la;
    w.okey_b = (unsigned int *)w.lb;
    w.ids_in = w.rank;
    w.d0 = carve((size_t)n);
    w.d1 = carve((size_t)n);
    w.sel = carve(select_workspace_bytes());
    w.median = (double *)carve(16);
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
// The whole segmentation, enqueued on `st`.  presorted: the neighbour search ran in POSITION mode (knn.hip) and left, inside this
// workspace, the cloud and its normals in the search's own cell order (w.xyz_p, w.nrm_p), the caller's index of every position
// (w.orig) and the TRANSPOSED neighbour lists in position space (w.knnT): the segmentation works in that order as it stands --
// no ordering sort, no gather of the cloud, no transpose (rounds 3-4 paid 2.3 ms and 8.5 GB per 10 M points for the three).
static int segment_enqueue(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k, double resolution,
                           const float *grid_bbox_host, int32_t *labels_out, int32_t *reps_out, int32_t *info_out, f4l::svg::Ws &w,
                           hipStream_t st, bool presorted) {
    using namespace f4l;
    using namespace f4l::svg;
    int rc = F4L_OK;
    GridBox box;
    box.given = grid_bbox_host ? 1 : 0;
    for (int d = 0; d < 3; ++d) {
        box.mn[d] = grid_bbox_host ? grid_bbox_host[d] : 0.f;
        box.mx[d] = grid_bbox_host ? grid_bbox_host[3 + d] : 0.f;
        if (grid_bbox_host && !(box.mx[d] >= box.mn[d])) return F4L_EINVAL;
    }
    const dim3 g(GRID), ga(GRID_ACTIVE), b(BLOCK), one(1);

    F4L_HIP_CHECK(hipMemsetAsync(w.st, 0, sizeof(State), st));
    hipLaunchKernelGGL(init_state_kernel, one, one, 0, st, w.st, (int32_t)n, box);
    size_t tb = w.prim_bytes;
    if (presorted) {
        xyz = w.xyz_p;
        normals = w.nrm_p;
        hipLaunchKernelGGL(svg::bbox_kernel, dim3(256), b, 0, st, xyz, n, w.st, box.given);
    } else {
        // K
        hipLaunchKernelGGL(svg::bbox_kernel, dim3(256), b, 0, st, xyz, n, w.st, box.given);
        // the segmentation's own order of the points (F4L_SV_NO_REORDER: the caller's, for measurements)
        hipLaunchKernelGGL(order_key_kernel, g, b, 0, st, xyz, n, resolution, w.st, getenv("F4L_SV_NO_REORDER") ? 1 : 0, w.okey_a, w.ids_in);
        F4L_LAUNCH_CHECK();
        F4L_HIP_CHECK(rocprim::radix_sort_pairs(w.prim, tb, w.okey_a, w.okey_b, w.ids_in, w.orig, (size_t)n, 0, 32, st, false));
        hipLaunchKernelGGL(order_apply_kernel, g, b, 0, st, xyz, normals, w.orig, n, w.xyz_p, w.nrm_p, w.pos_of);
        xyz = w.xyz_p;  // (from here on: the sorted copies)
        normals = w.nrm_p;
    }
    {   // K: the distinct cells of the resolution grid, through a hash set in the (still idle) filter table: two slots per point
        const unsigned long long slots = 2ULL * (unsigned long long)n + 1024ULL;  // (layout: the table holds at least that many words)
        F4L_HIP_CHECK(hipMemsetAsync(w.keys_a, 0xff, (size_t)slots * 8, st));
        hipLaunchKernelGGL(grid_count_kernel, g, b, 0, st, xyz, n, resolution, w.st, w.keys_a, slots);
        F4L_LAUNCH_CHECK();
    }
    // lambda0
    if (!presorted) hipLaunchKernelGGL(knn_transpose_kernel, g, b, 0, st, knn, w.orig, w.pos_of, n, k, w.knnT);
    if (presorted) hipLaunchKernelGGL(min_metric_kernel<true>, g, b, 0, st, xyz, normals, w.knnT, n, k, resolution, w.dis, w.st);
    else hipLaunchKernelGGL(min_metric_kernel<false>, g, b, 0, st, xyz, normals, w.knnT, n, k, resolution, w.dis, w.st);
    F4L_LAUNCH_CHECK();
    {   // the median of the smallest neighbour metrics: one order statistic, no sort
        const int64_t rank = n / 2;
        rc = select_ranks_f64(w.dis, n, 1, 1, &rank, w.median, w.sel, st);
        if (rc != F4L_OK) return rc;
    }
    hipLaunchKernelGGL(start_kernel, one, one, 0, st, w.st, (const double *)w.median);
    hipLaunchKernelGGL(init_points_kernel, g, b, 0, st, n, w.parent, w.size, w.bestm, w.bestu);
    F4L_LAUNCH_CHECK();
    // fusion, labels and the exchange: the schedule of launches clouds normally need, then one-workgroup kernels for
    // whatever is left (see segment_rest_kernel).  F4L_SV_LAUNCHES=1: the whole schedule as launches, whose tail returns at
    // once.  F4L_SV_SCHEDULED="rounds,sweeps" overrides the split (tests run "2,1" and "0,0": nearly everything in the
    // one-workgroup kernels).
    int sched_rounds = SCHED_ROUNDS, sched_sweeps = SCHED_SWEEPS;
    if (getenv("F4L_SV_LAUNCHES")) { sched_rounds = LAMBDA_ROUNDS; sched_sweeps = SWEEPS; }
    if (const char *e = getenv("F4L_SV_SCHEDULED")) {
        int a = 0, c = 0;
        if (sscanf(e, "%d,%d", &a, &c) == 2 && a >= 0 && a <= LAMBDA_ROUNDS && c >= 0 && c <= SWEEPS) { sched_rounds = a; sched_sweeps = c; }
    }
    SegArgs sa;
    sa.xyz = xyz; sa.nrm = normals; sa.orig = w.orig; sa.knnT = w.knnT; sa.n = n; sa.k = k; sa.resolution = resolution; sa.st = w.st;
    sa.edges_a = w.edges_a; sa.edges_b = w.edges_b; sa.table = w.table; sa.akey = w.akey; sa.am = w.am; sa.bestm = w.bestm;
    sa.prop_key = w.prop_key; sa.parent = w.parent; sa.size = w.size; sa.bestu = w.bestu; sa.prop_u = w.prop_u; sa.prop_v = w.flag; sa.la = w.la;
    sa.lb = w.lb; sa.dis = w.dis; sa.d0 = w.d0; sa.d1 = w.d1; sa.node = w.node;
    sa.reps_a = w.prop_u; sa.reps_b = w.rank;
    // the device-wide filter of parallel edges: cleared ONCE -- every entry carries the number of the round that wrote it (round_stamp)
    F4L_HIP_CHECK(hipMemsetAsync(w.table, 0xff, (size_t)table_size((unsigned long long)n * (unsigned long long)k) * 8, st));
    for (int r = 0; r < sched_rounds; ++r) {
        hipLaunchKernelGGL(node_kernel, g, b, 0, st, sa, r);
        if (r <= KNN_ROUNDS) hipLaunchKernelGGL(rows_kernel, g, b, 0, st, sa, r);
        else hipLaunchKernelGGL(build_kernel, g, dim3(BUILD_BLOCK), 0, st, sa, r);
        const dim3 gs = r < SVX_WIDE_ROUNDS ? g : ga;  // (the first rounds' lists are millions of edges: every pass is a handful of dependent loads per edge, and more threads in flight are fewer trips)
        for (int s = 0; s < SUBROUNDS; ++s) {
            hipLaunchKernelGGL(cand_kernel, gs, b, 0, st, sa, r, s);
            hipLaunchKernelGGL(cand2_kernel, gs, b, 0, st, sa, r, s);
            hipLaunchKernelGGL(apply_kernel, gs, b, 0, st, sa, r, s);
        }
        F4L_LAUNCH_CHECK();
    }
    if (sched_rounds < LAMBDA_ROUNDS) hipLaunchKernelGGL(segment_rest_kernel, one, dim3(1024), 0, st, sa, sched_rounds, 0);
    hipLaunchKernelGGL(overflow_collect_kernel, dim3(256), dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(overflow_select_kernel, one, dim3(1024), 0, st, w.st, w.prop_key);
    hipLaunchKernelGGL(overflow_apply_kernel, g, b, 0, st, w.st, w.prop_key, w.prop_u, w.flag, w.parent, w.size);  // (prop_v lives in `flag`, idle until the relabelling)
    // labels and the boundary exchange
    hipLaunchKernelGGL(flatten_kernel, g, b, 0, st, n, w.parent);
    hipLaunchKernelGGL(labels_init_kernel, g, b, 0, st, xyz, normals, resolution, n, w.parent, w.la, w.lb, w.dis);
    F4L_HIP_CHECK(hipMemsetAsync(w.d0, 0, (size_t)n, st));
    F4L_HIP_CHECK(hipMemsetAsync(w.d1, 0, (size_t)n, st));
    for (int s = 0; s < sched_sweeps; ++s) hipLaunchKernelGGL(sweep_kernel, g, b, 0, st, sa, s);
    if (sched_sweeps < SWEEPS) hipLaunchKernelGGL(segment_rest_kernel, one, dim3(1024), 0, st, sa, -1, sched_sweeps);
    F4L_LAUNCH_CHECK();
    // relabel 0..K-1 in ascending order of the representative's index
    hipLaunchKernelGGL(root_flag_kernel, g, b, 0, st, n, w.parent, w.orig, w.flag);
    F4L_LAUNCH_CHECK();
    tb = w.prim_bytes;
    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim, tb, w.flag, w.rank, 0, (size_t)n, rocprim::plus<int32_t>(), st, false));
    hipLaunchKernelGGL(relabel_kernel, g, b, 0, st, w.st, n, w.la, w.lb, w.orig, w.flag, w.rank, labels_out, reps_out, info_out);
    F4L_LAUNCH_CHECK();
    return rc;
}

extern "C" int f4l_supervoxel_segment_device(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k,
                                             double resolution, const float *grid_bbox_host, int32_t *labels_out,
                                             int32_t *reps_out, int32_t *info_out, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    using namespace f4l;
    using namespace f4l::svg;
    if (!xyz || !normals || !knn || n <= 0 || k < 1 || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n >= MAX_PLACES) return F4L_EUNSUPPORTED;  // (the transpose tile holds rows of up to 64 neighbours; edge keys keep 28 bits per end)
    for (int d = 0; d < 3 && grid_bbox_host; ++d)
        if (!(grid_bbox_host[3 + d] >= grid_bbox_host[d])) return F4L_EINVAL;
    Ws w;
    const int rc = layout(n, k, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    return segment_enqueue(xyz, normals, knn, n, k, resolution, grid_bbox_host, labels_out, reps_out, info_out, w, (hipStream_t)stream, false);
}

// ---- the partition in two calls: neighbours (position mode), then the segmentation on what they left in the workspace -----------
// The neighbour search does not depend on the resolution: a caller that derives the resolution from the point spacing
// (src/coarse_to_fine_matching_base.py:2668-2671: sqrt(3) * 10 * median spacing) runs it first, takes the nearest-neighbour
// distances from it (nn1_d2_out, position order: good for a median), and then segments.  The search's scratch lives in the
// segmentation's edge buffers (idle until the fourth round), or behind the segmentation's workspace where those are too small.
namespace f4l {
int knn_position_mode(const float *xyz, int64_t n, int k, int32_t *knnT_out, double *normals_p_out, double *nn1_p_out, float *xyz_p_out,
                      int32_t *orig_out, void *workspace, size_t workspace_bytes, void *stream);
namespace svg {
static size_t partition_knn_offset(int64_t n, int k, const Ws &w, size_t &total) {
    const size_t need = f4l_knn_workspace_bytes(n, k);
    const size_t room = 2 * align_up((size_t)n * (size_t)k * 8);  // edges_a + edges_b, carved one behind the other
    if (need <= room) { total = w.total; return (size_t)((unsigned char *)w.edges_a - (unsigned char *)w.st); }
    total = w.total + need;
    return w.total;
}
}  // namespace svg
}  // namespace f4l
extern "C" size_t f4l_partition_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    f4l::svg::Ws w;
    if (f4l::svg::layout(n, k, w, (unsigned char *)256) != F4L_OK) return 0;  // (a non-null base: the offsets are wanted)
    size_t total = 0;
    (void)f4l::svg::partition_knn_offset(n, k, w, total);
    return total;
}
extern "C" int f4l_partition_neighbours(const float *xyz, int64_t n, int k, double *nn1_d2_out, void *workspace, size_t workspace_bytes,
                                        void *stream) {
    using namespace f4l::svg;
    if (!xyz || n <= 0 || k < 1 || k >= n || !workspace) return F4L_EINVAL;  // supervoxel.cpp:100
    if (k > F4L_MAX_K || n >= MAX_PLACES) return F4L_EUNSUPPORTED;
    Ws w;
    int rc = layout(n, k, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    size_t total = 0;
    const size_t at = partition_knn_offset(n, k, w, total);
    if (workspace_bytes < total) return F4L_EWORKSPACE;
    return f4l::knn_position_mode(xyz, n, k, w.knnT, w.nrm_p, nn1_d2_out, w.xyz_p, w.orig, (unsigned char *)workspace + at,
                                  f4l_knn_workspace_bytes(n, k), stream);
}
extern "C" int f4l_partition_segment(int64_t n, int k, double resolution, const float *grid_bbox_host, int32_t *labels_out,
                                     int32_t *reps_out, int32_t *info_out, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace f4l::svg;
    if (n <= 0 || k < 1 || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n >= MAX_PLACES) return F4L_EUNSUPPORTED;
    for (int d = 0; d < 3 && grid_bbox_host; ++d)
        if (!(grid_bbox_host[3 + d] >= grid_bbox_host[d])) return F4L_EINVAL;
    Ws w;
    const int rc = layout(n, k, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    return segment_enqueue(nullptr, nullptr, nullptr, n, k, resolution, grid_bbox_host, labels_out, reps_out, info_out, w, (hipStream_t)stream, true);
}

extern "C" size_t f4l_supervoxel_parallel_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    const size_t a = f4l_knn_workspace_bytes(n, k), s = f4l_supervoxel_segment_device_workspace_bytes(n, k);
    const size_t idx = ((size_t)n * k * 4 + 255) / 256 * 256, nrm = ((size_t)n * 24 + 255) / 256 * 256;
    const size_t two_step = (a > s ? a : s) + idx + nrm;  // the kNN workspace is dead when the segmentation starts
    const size_t fused = f4l_partition_workspace_bytes(n, k);
    return two_step > fused ? two_step : fused;
}

// kNN + normals + segmentation, all on the device.  f4l_knn synchronises `stream` once while it sizes its grid (bounding
// box and cell count are read back); nothing after that does.  Without a caller who wants the neighbour lists or the normals
// (caller's order) the search runs in position mode and the segmentation takes its order over (f4l_partition_neighbours +
// f4l_partition_segment); with one, or for k beyond the lane-per-query search, the two stages run as before.
extern "C" int f4l_supervoxel_parallel(const float *xyz, int64_t n, int k, double resolution, int32_t *labels_out,
                                       int32_t *reps_out, int32_t *info_out, int32_t *knn_out, double *normals_out,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    if (!xyz || n <= 0 || k < 1 || k >= n || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;  // supervoxel.cpp:100
    if (k > F4L_MAX_K || n >= f4l::svg::MAX_PLACES) return F4L_EUNSUPPORTED;
    if (workspace_bytes < f4l_supervoxel_parallel_workspace_bytes(n, k)) return F4L_EWORKSPACE;
    if (!knn_out && !normals_out && !getenv("F4L_SV_TWO_STEP")) {
        const int rc = f4l_partition_neighbours(xyz, n, k, nullptr, workspace, workspace_bytes, stream);
        if (rc == F4L_OK) return f4l_partition_segment(n, k, resolution, nullptr, labels_out, reps_out, info_out, workspace, workspace_bytes, stream);
        if (rc != F4L_EUNSUPPORTED) return rc;
    }
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
