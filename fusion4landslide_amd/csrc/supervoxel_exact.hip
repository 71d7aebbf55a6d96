// supervoxel_exact.hip -- the reference's OWN supervoxel labels on the device.
//
// codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-248 is sequential twice over: the fusion (:117-176) visits the
// representatives in index order, each absorbing what its closure offers under the state its predecessors left (mutable ordered
// adjacency lists, a disjoint set, a `break` when the count reaches K), and the boundary exchange (:186-237) is a FIFO work list.
// Rounds 1-4 replayed both on one host core (csrc/supervoxel_host.cpp: 1.07 s per 1 M points) and offered a parallel VARIANT
// with other labels (csrc/supervoxel_gpu.hip).  This file computes the sequential result itself, in parallel, as a FIXED POINT:
//
//   fusion    What centre i does is a function of the round's starting state and of what the centres BEFORE it (lower index) did.
//             Every centre is evaluated at once against an ESTIMATE of that -- per node: who absorbed it (abs), and per centre:
//             its size and its ordered list of rejected neighbours after its turn (ns, kept) -- seen through the eyes of centre i:
//             an absorption by centre c counts iff c < i and c itself was not absorbed before its turn (abs[c] < c); a root j has
//             its new size and list iff j < i.  The evaluations give the next estimate; centre i is right as soon as all centres
//             below it are, so the iteration reaches THE sequential result (the unique fixed point of a recursion on the index)
//             after as many passes as the longest chain of real dependencies: 6-20 per lambda round on 100 k - 1 M point clouds
//             in patch, row-major and random order (tools/experiments/fixed_point_fusion_proto.cpp, every label equal).  The
//             `break` at K representatives is one more dependency on lower centres: a centre's budget is what the absorptions of
//             the centres before it (a prefix sum of the estimate) leave of (live - K).
//   exchange  The queue is processed in GENERATIONS (the entries it holds when a generation starts; what they push is the next
//             generation, in (position of the pusher, neighbour slot) order).  Inside a generation an entry sees the labels its
//             EARLIER entries leave: the same fixed-point iteration, 3-4 generations and ~10 passes in all.
//
// Queue order, visited sets, list order, the strict `<` of the exchange and the position of the break are the reference's, so the
// labels are the reference's (tests: every label of the reference-compiled fixtures and of fresh clouds through the live
// reference, and of the host replay on random clouds).  One wavefront evaluates one centre (its queue in LDS).  A closure beyond
// the queue's capacity, or lists beyond the pools, are reported (F4L_EUNSUPPORTED): f4l_supervoxel then replays on the host.
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "select.h"
#include "sv_metric.h"

namespace f4l {
namespace svx {

constexpr unsigned int NONE = 0xffffffffu;
constexpr int QCAP_WIDE = 1024;                 // distinct nodes one centre's closure may visit (its queue, in LDS)
#ifndef SVX_EVAL_WAVES
#define SVX_EVAL_WAVES 4
#endif
constexpr int EVAL_WAVES = SVX_EVAL_WAVES;  // wavefronts per workgroup
constexpr int SUBPOOLS = 1024;            // bump pointers of a list pool (one address would serialise a million allocations)
constexpr unsigned long long KEY_INF = (1ULL << 40) - 1ULL;
constexpr int32_t POS_INF = 0x7fffffff;
constexpr int MAX_ROUNDS = 64, MAX_ITERS = 4096, MAX_GENERATIONS = 100000;

struct State {
    unsigned int bb[6];
    int K;
    int overflow;          // 1: a closure beyond QCAP, 2: a list pool full
    int changed[8];        // per pass (slot = pass % 8): something differed from the estimate before
    int m;                 // entries of the exchange's current generation
    int max_tail;          // the largest closure (visited set) any centre had
    int done, done_it;     // the round's estimate has converged (abs_changed_kernel), after this many passes: the passes queued behind it do nothing
    unsigned int arrive;   // workgroups of abs_changed_kernel that are through (the last one closes the pass)
    long long total[8];    // per pass: absorptions of the whole estimate (the budget of the last round is switched on by it)
    unsigned long long pool_max, pool_sum;  // (statistics: the fullest slice of a list pool any pass left, and the most entries a pass wrote)
#ifdef SVX_MEASURE_PREFIX
    int minchg[8], nchg[8];  // (measurement build: the lowest centre whose output changed in a pass, and how many did)
#endif
    unsigned long long sub[2][SUBPOOLS];  // bump pointers of the two estimate pools
};

__device__ __forceinline__ unsigned int f2ord(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
#define SVX_FOR(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)
// The same for kernels that GATHER around the points they own (neighbours, representatives).  Workgroups are dealt to the 8 XCDs
// round robin and an XCD's L2 is its own: under the plain grid stride every XCD sees every eighth chunk of the window the grid
// is in -- 0.5 M points, 33 MB of node records at 64 bytes -- and their neighbours all over it, eight times what its 4 MB hold.
// Here an XCD walks one contiguous eighth of the range with its own workgroups side by side in it (a window of 64 k points),
// so a record fetched for one point is still there for the thirty that share it (round 6; gridDim.x: a multiple of 8).
#ifndef SVX_NO_XCD_MAP
__device__ __forceinline__ bool xcd_grid() { return (gridDim.x & 7u) == 0u; }  // (else -- a grid below 8 or ragged -- the plain stride)
#define SVX_FOR_XCD(i, n)                                                                                                                  \
    for (int64_t i##_x = xcd_grid() ? 1 : 0,                                                                                               \
                 i##_per = i##_x ? ((((int64_t)(n) + 7) >> 3) + blockDim.x - 1) / blockDim.x * blockDim.x : (int64_t)(n),                   \
                 i##_lo = i##_x ? (int64_t)(blockIdx.x & 7u) * i##_per : 0,                                                                \
                 i##_hi = i##_lo + i##_per < (int64_t)(n) ? i##_lo + i##_per : (int64_t)(n),                                               \
                 i##_st = (int64_t)(i##_x ? gridDim.x >> 3 : gridDim.x) * blockDim.x,                                                      \
                 i = i##_lo + (int64_t)(i##_x ? blockIdx.x >> 3 : blockIdx.x) * blockDim.x + threadIdx.x;                                  \
         i < i##_hi; i += i##_st)
#else
__device__ __forceinline__ bool xcd_grid() { return false; }
#define SVX_FOR_XCD(i, n) SVX_FOR(i, n)
#endif
// ... and with a DPP row of 16 lanes per item (blockDim.x / 16 items per workgroup and trip): 16 gathers in flight per item where a
// thread per item walks its neighbours one dependent load after the other
#define SVX_FOR_XCD_ROWS(i, n)                                                                                                             \
    for (int64_t i##_x = xcd_grid() ? 1 : 0, i##_r = (int64_t)(blockDim.x >> 4),                                                           \
                 i##_per = i##_x ? ((((int64_t)(n) + 7) >> 3) + i##_r - 1) / i##_r * i##_r : (int64_t)(n),                                  \
                 i##_lo = i##_x ? (int64_t)(blockIdx.x & 7u) * i##_per : 0,                                                                \
                 i##_hi = i##_lo + i##_per < (int64_t)(n) ? i##_lo + i##_per : (int64_t)(n),                                               \
                 i##_st = (int64_t)(i##_x ? gridDim.x >> 3 : gridDim.x) * i##_r,                                                           \
                 i = i##_lo + (int64_t)(i##_x ? blockIdx.x >> 3 : blockIdx.x) * i##_r + (threadIdx.x >> 4);                                \
         i < i##_hi; i += i##_st)

// ---- K = occupied cells of the resolution grid (grid_sample.h:48-68), lambda0's metric sweep ------------------------------------
__global__ void bbox_kernel(const float *__restrict__ xyz, int64_t n, State *st) {
    unsigned int mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    SVX_FOR(i, n) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned int o = f2ord(xyz[3 * i + d]);
            mn[d] = o < mn[d] ? o : mn[d];
            mx[d] = o > mx[d] ? o : mx[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int a = (unsigned int)__shfl_xor((int)mn[d], m, 64), b = (unsigned int)__shfl_xor((int)mx[d], m, 64);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b > mx[d] ? b : mx[d];
        }
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomicMin(&st->bb[d], mn[d]); atomicMax(&st->bb[3 + d], mx[d]); }
    }
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
    SVX_FOR(i, n) {
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)(((double)xyz[3 * i + d] - mn[d]) / resolution);
            c[d] = c[d] < 0 ? 0 : (c[d] > size[d] - 1 ? size[d] - 1 : c[d]);
        }
        const unsigned long long key = ((unsigned long long)c[0] * (unsigned long long)size[1] + (unsigned long long)c[1]) * (unsigned long long)size[2] +
                                       (unsigned long long)c[2];
        unsigned long long slot = __umul64hi(mix64(key), slots);
        for (;;) {
            const unsigned long long old = atomicCAS(&set[slot], ~0ULL, key);
            if (old == ~0ULL) { ++cnt; break; }
            if (old == key) break;
            slot = slot + 1ULL == slots ? 0ULL : slot + 1ULL;
        }
    }
    cnt = wave_sum(cnt);
    if (lane_id() == 0 && cnt) atomicAdd(&st->K, cnt);
}
// ---- fusion -------------------------------------------------------------------------------------------------------------
// What an evaluation gathers lives in two packed records per node (round 6; before: nine separate arrays, i.e. six to seven 32-byte
// sectors per visited node and two per list entry -- the passes are bound by exactly those gathers):
//   NodeS  (64 B, one cache line half)  what a ROUND starts from: position, normal, size, list.  Written by iota / commit.
//   NodeE  (32 B, one sector)           one ESTIMATE of "what the centres of this round do", two of them (read / written): the
//                                       node's own outcome as a centre -- size and kept list after its turn, the nodes it absorbs.
//   abs    (4 B, dense)                 ... and who absorbs the node (claims by atomicMin, from any centre): what every list entry's
//                                       chain walk reads, sixteen nodes to a sector (inside the records it cost a sector each and
//                                       the pass's comparison read 64 bytes per centre for 8).
// The lists (neighbour table, then the pools) hold ROUND-START ROOTS: rootlists_kernel maps the survivors' lists through `root`
// once per round, instead of every evaluation of every pass doing it per entry.
struct alignas(64) NodeS {
    float x, y, z;
    int32_t sz;                  // round-start size
    double nx, ny, nz;
    int64_t off;                 // round-start list: offset into `lists` ...
    int32_t len;                 //                   ... and length
    int32_t pad_[3];
};
// (measured and not kept, round 6: the record as two arrays, {x, y, z, size} 16 B and {normal, offset | length} 32 B, so that more
//  neighbours share a cache line: 228 against 215 ms per 10 M points -- the second gather per visited node costs more than the lines save)
struct alignas(32) NodeE {
    int32_t pad0_;
    int32_t ns, cnt, len;        // as a centre: its size after its turn, the nodes it absorbs, its kept list's length
    int64_t off;                 //              ... and offset
    unsigned long long hash;     // order-sensitive hash of that list (what "the list did not change" is read from)
};
static_assert(sizeof(NodeS) == 64 && sizeof(NodeE) == 32, "record sizes");
__device__ __forceinline__ NodeS load_s(const NodeS *p) {
    NodeS r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 *o = reinterpret_cast<uint4 *>(&r);
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    return r;
}
__device__ __forceinline__ NodeE load_e(const NodeE *p) {
    NodeE r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 *o = reinterpret_cast<uint4 *>(&r);
    o[0] = q[0]; o[1] = q[1];
    return r;
}
struct FuseArgs {
    NodeS *S;                    // [n]
    const int32_t *lists;        // the neighbour table followed by the three list pools: one address space
    int64_t n;
    int k;
    double resolution;
    State *st;
    const int32_t *reps;         // the round's centres, ascending
    int nreps;
    const int64_t *before;       // [nreps] absorptions by the centres before each centre (previous estimate), or null
    int64_t pool_base, sub_cap;  // where this pass writes lists: SUBPOOLS regions of sub_cap entries from pool_base
    int pool_sel;                // which set of bump pointers
    NodeE *rd, *wr;              // [n] the estimate read / written
    unsigned int *abs_rd, *abs_wr;  // [n] its claims
    int32_t *cnt_slot_rd, *cnt_slot_wr;  // [nreps] cnt in the order of the centres (the budget's prefix sum runs over it)
};

__global__ void iota_kernel(int32_t *root, int32_t *reps, NodeS *S, const float *__restrict__ xyz, const double *__restrict__ nrm, int64_t n, int k) {
    SVX_FOR(i, n) {
        root[i] = (int32_t)i; reps[i] = (int32_t)i;
        NodeS r;
        r.x = xyz[3 * i]; r.y = xyz[3 * i + 1]; r.z = xyz[3 * i + 2];
        r.sz = 1;
        r.nx = nrm[3 * i]; r.ny = nrm[3 * i + 1]; r.nz = nrm[3 * i + 2];
        r.off = i * k; r.len = k;
        r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
        const uint4 *o = reinterpret_cast<const uint4 *>(&r);
        uint4 *q = reinterpret_cast<uint4 *>(S + i);
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = o[3];
    }
}
// lambda0's sweep (:105-113): every point's smallest metric to a neighbour, from the packed records (one line per neighbour instead of
// the coordinate array's and the normal array's)
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void min_metric_kernel(const NodeS *__restrict__ S, const int32_t *__restrict__ knn, int64_t n, int k, double resolution, double *__restrict__ dis0) {
    const int sub = (int)(threadIdx.x & 15);
    SVX_FOR_XCD_ROWS(i, n) {  // (16 lanes per point: the minimum of a set does not depend on the order it is taken in)
        const NodeS si = load_s(S + i);
        const float pi_[3] = {si.x, si.y, si.z};
        const double ni_[3] = {si.nx, si.ny, si.nz};
        double best = DBL_MAX;
        for (int j = sub; j < k; j += 16) {
            const int64_t q = knn[i * k + j];
            if (q != i) {
                const NodeS sq = load_s(S + q);
                const float pq[3] = {sq.x, sq.y, sq.z};
                const double nq[3] = {sq.nx, sq.ny, sq.nz};
                const double m = sv_metric_vals(pi_, ni_, pq, nq, resolution);
                best = m < best ? m : best;
            }
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
            const double o = __shfl_xor(best, x, 64);
            best = o < best ? o : best;
        }
        if (sub == 0) dis0[i] = best;
    }
}
// (and what pass 0 needs prepared: its pool's bump pointers, its claims, its flag and its total -- every later pass is prepared by
//  the last workgroup of the pass before it, abs_changed_kernel)
__global__ void round_init_kernel(FuseArgs a) {
    if (blockIdx.x == 0) {
        for (int t = (int)threadIdx.x; t < SUBPOOLS; t += (int)blockDim.x) a.st->sub[0][t] = 0ULL;
        if (threadIdx.x == 0) {
            a.st->done = 0; a.st->done_it = 0; a.st->arrive = 0u; a.st->changed[0] = 0; a.st->total[0] = 0;
#ifdef SVX_MEASURE_PREFIX
            a.st->minchg[0] = 0x7fffffff; a.st->nchg[0] = 0;
#endif
        }
    }
    SVX_FOR(s, a.nreps) {
        const int32_t i = a.reps[s];
        const NodeS si = load_s(a.S + i);
        a.abs_wr[i] = NONE;
        a.abs_rd[i] = NONE;
        NodeE e;
        e.pad0_ = 0; e.ns = si.sz; e.cnt = 0; e.len = si.len; e.off = si.off;
        e.hash = ~0ULL;   // ("the round-start list": no evaluation writes this value twice in a row unless nothing changes)
        const uint4 *o = reinterpret_cast<const uint4 *>(&e);
        uint4 *q = reinterpret_cast<uint4 *>(a.rd + i);
        q[0] = o[0]; q[1] = o[1];
        a.cnt_slot_rd[s] = 0;
    }
}
// Half a wavefront (32 lanes) = one centre: the lists are a few dozen entries, and two dependent-load chains per wave keep the
// memory system busier than one.  Q: the closure's queue = its visited set, in the reference's order (:125-134, 151-157), in LDS.
// Everything "uniform" below is uniform within a half; both halves walk every loop together (a half that is through is
// predicated off), so that the wave-wide ballots and shuffles stay convergent.
//
// (Measured and dropped: skipping centres none of whose inputs changed in the pass before -- per-node change stamps, the nodes a
//  centre read stored behind its list.  A change at one node sends its ~30 neighbouring centres back into evaluation whether
//  or not their outcome moves, so fewer than half are skipped until the last passes, and the bookkeeping costs every evaluation:
//  51.4 ms of passes per 1 M points against 41.7 without.  One centre per wavefront, the first version: 41.7 ms.)
#ifndef SVX_EVAL_WPE
#define SVX_EVAL_WPE 6  // (round 6: 80 VGPRs and no spill beat 64 with; 6 waves per SIMD are what the LDS leaves anyway)
#endif
// G lanes per centre (64 / G centres per wavefront), QCAP entries of queue: <16, 256> is the shape clouds normally take (closures
// of 45-76 nodes on surfaces, up to 216 in volumes; 16 KB of LDS per workgroup: full occupancy; 31 ms per 1 M points against 37
// for <32, 512>); a closure beyond it restarts the segmentation in the wide shape <32, 1024>, one beyond that replays on the host.
template <int G, int QCAP>
__global__ __launch_bounds__(EVAL_WAVES * 64, SVX_EVAL_WPE) void eval_kernel(FuseArgs a, double lambda, long long budget_total, int pass) {
    constexpr int CPW = 64 / G;  // centres per wavefront
    constexpr unsigned int GMASK = G == 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u);
    __shared__ int32_t q_all[EVAL_WAVES][CPW][QCAP];
    __shared__ unsigned int acc_all[EVAL_WAVES][CPW][QCAP / 32];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id(), h = lane / G, hl = lane % G, hbase = h * G;
    const int64_t s0 = ((int64_t)blockIdx.x * EVAL_WAVES + wave) * CPW;
    if (s0 >= a.nreps) return;  // (whole wave)
    if (a.st->done) return;     // (converged earlier in this batch of passes)
    const int64_t s = s0 + h;
    const bool valid_c = s < a.nreps;
    int32_t *Q = q_all[wave][h];
    unsigned int *ACC = acc_all[wave][h];
    const int32_t i = a.reps[valid_c ? s : s0];
    const unsigned int ui = (unsigned int)i;
    const NodeE *__restrict__ E = a.rd;
    const unsigned int *__restrict__ ABS = a.abs_rd;
    int32_t *__restrict__ pool = const_cast<int32_t *>(a.lists);
    const unsigned int below = (1u << hl) - 1u;
    auto hb = [&](unsigned long long m) { return (unsigned int)(m >> hbase) & GMASK; };  // this half's bits of a wave-wide ballot
    auto half_sum = [&](int v) {
#pragma unroll
        for (int x = 1; x < G; x <<= 1) v += __shfl_xor(v, x, 64);
        return v;
    };
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const NodeE ei = load_e(E + i);    // (its claim, and what it did in the pass before: one sector)
    const NodeS si = load_s(a.S + i);  // (its position and normal -- the metric's first argument, :142 --, its size and list: one line)
    const unsigned int absi = a.abs_rd[i];
    const bool dead = absi != NONE && absi < ui;  // absorbed before its turn: adjacents[i] is empty by then (:121)
    const int64_t off0 = si.off;
    const int len0 = si.len;
    long long budget = a.before ? budget_total - (long long)a.before[valid_c ? s : s0] : 0x7fffffffffffLL;
    const bool run = valid_c && !dead && len0 > 0 && budget > 0;
    int nsz = si.sz, cnt = 0;
    int head = 0, tail = 0;
    bool ovf = false;
    const float pi_[3] = {si.x, si.y, si.z};
    const double ni_[3] = {si.nx, si.ny, si.nz};

    // appends Find(list entries) that are not yet visited, in list order (:126-133 / :151-157); `act`: this half has a list to append
    auto append_list = [&](bool act, int64_t off, int len) {
        for (int c0 = 0; __any(act && !ovf && c0 < len); c0 += G) {
            const bool go = act && !ovf && c0 < len;
            const bool have = go && c0 + hl < len;
            unsigned int r = NONE;
            if (have) {
                r = (unsigned int)pool[off + c0 + hl];  // (a round-start root: rootlists_kernel)
                // Find as centre i sees it (set.Find, :127/:152): every absorption by a centre that ran BEFORE i and was alive at its
                // own turn; honoured claims lead to ever higher centres, so the walk ends
                for (;;) {
                    const unsigned int c = ABS[r];
                    if (c == NONE || !(c < ui)) break;
                    const unsigned int cc = ABS[c];
                    if (cc != NONE && cc < c) break;
                    r = c;
                }
            }
            // visited already? (the queue so far; four entries per LDS read; entries past `tail` hold -1)
            bool fresh = have;
            for (int e = 0; __any(go && e < tail); e += 4) {
                if (go && e < tail) {
                    const int4 v = *reinterpret_cast<const int4 *>(Q + e);
                    fresh = fresh && (unsigned int)v.x != r && (unsigned int)v.y != r && (unsigned int)v.z != r && (unsigned int)v.w != r;
                }
            }
            // ... or by a lower lane of this chunk (the first occurrence wins)
            unsigned int cand = hb(__ballot(fresh)), news = 0u;
            while (__any(cand != 0u)) {
                const bool on = cand != 0u;
                const int l = on ? __ffs((int)cand) - 1 : 0;
                const unsigned int rl = (unsigned int)__shfl((int)r, hbase + l, 64);
                const unsigned int same = hb(__ballot(on && fresh && r == rl));
                if (on) { news |= 1u << l; cand &= ~same; }
            }
            if (go) {
                const int nn = (int)__popc(news);
                if (tail + nn + 4 > QCAP) ovf = true;
                else {
                    if ((news >> hl) & 1u) Q[tail + (int)__popc(news & below)] = (int32_t)r;
                    if (hl < 4) Q[tail + nn + hl] = -1;
                    tail += nn;
                }
            }
            wsync();
        }
    };

    if (run) {
        if (hl < 5) Q[hl] = hl == 0 ? i : -1;  // visited[i] = true; queue[front++] = i (:123-125)
        for (int e = hl; e < QCAP / 32; e += G) ACC[e] = 0u;
        head = tail = 1;
    }
    wsync();
    append_list(run, off0, len0);
    bool stop = false;
    while (__any(run && !stop && !ovf && head < tail)) {  // :137-163, 32 entries of the queue at a time
        const bool go = run && !stop && !ovf && head < tail;
        const int m = go ? (tail - head < G ? tail - head : G) : 0;
        const bool mine = hl < m;
        const int32_t j = mine ? Q[head + hl] : i;
        // sizes[j] and adjacents[j] as centre i finds them (:142, :151): j ran before i -- grew, kept a list of its own -- iff j < i
        int sj = 0, jlen = 0;
        int64_t joff = 0;
        bool acc = false;
        if (mine) {
            const NodeS sn = load_s(a.S + j);
            sj = sn.sz; joff = sn.off; jlen = sn.len;
            if ((unsigned int)j < ui) {
                const NodeE en = load_e(E + j);  // (the sector its claim was read from a moment ago)
                sj = en.ns; joff = en.off; jlen = en.len;
            }
            const float pj[3] = {sn.x, sn.y, sn.z};
            const double nj[3] = {sn.nx, sn.ny, sn.nz};
            const double loss = (double)sj * sv_metric_vals(pi_, ni_, pj, nj, a.resolution);  // :142 `sizes[j] * metric(points[i], points[j])`
            acc = lambda - loss > 0.0;                                                        // :143-144
        }
        unsigned int accm = hb(__ballot(acc));
        int done = m;  // entries of this chunk that the reference's loop reaches
        if (go && (long long)__popc(accm) >= budget) {  // `if (--number_of_supervoxels == n_supervoxels) break;` (:160) inside this chunk
            unsigned int t = accm;
            for (long long b = 1; b < budget; ++b) t &= t - 1u;
            const int last = __ffs((int)t) - 1;  // lane of the absorption that reaches K
            done = last + 1;
            accm &= (last == 31) ? ~0u : ((1u << (last + 1)) - 1u);
            stop = true;
        }
        const bool take = mine && ((accm >> hl) & 1u);
        if (take) {
            atomicMin(&a.abs_wr[j], ui);  // set.Link(j, i) (:145); of several centres that claim j in an estimate the lowest counts
            atomicOr(&ACC[(head + hl) >> 5], 1u << ((head + hl) & 31));
        }
        nsz += half_sum(take ? sj : 0);   // sizes[i] += sizes[j] (:147)
        const int na = (int)__popc(accm);
        cnt += na;
        budget -= na;
        // the absorbed nodes' lists join the queue, in the queue's order (:149-157)
        unsigned int mm = accm;
        while (__any(mm != 0u && !ovf)) {
            const bool ex = mm != 0u && !ovf;
            const int l = ex ? __ffs((int)mm) - 1 : 0;
            if (ex) mm &= mm - 1u;
            const long long o = __shfl((long long)joff, hbase + l, 64);
            const int ln = __shfl(jlen, hbase + l, 64);
            append_list(ex, (int64_t)o, ln);
        }
        if (go) head += done;
    }
    // the centre's list after its turn (:164 `adjacents[i].swap(adjacent)`): the entries it looked at and did not absorb, in order
    int64_t out_off = dead ? 0 : off0;
    int out_len = dead ? 0 : len0;  // (a centre that did not run keeps its list; an absorbed one's is cleared, :158)
    unsigned long long hash = dead ? 1ULL : ~0ULL;
    wsync();
    const bool fin = run && !ovf;
    int kept = 0;
    for (int e0 = 1; __any(fin && e0 < head); e0 += G) {
        const int e = e0 + hl;
        kept += (int)__popc(hb(__ballot(fin && e < head && !((ACC[e >> 5] >> (e & 31)) & 1u))));
    }
    if (fin) {
        out_len = kept;
        out_off = 0;
        if (kept > 0) {
            const int sp = (int)(s & (SUBPOOLS - 1));
            unsigned long long base = 0ULL;
            if (hl == 0) base = atomicAdd(&a.st->sub[a.pool_sel][sp], (unsigned long long)kept);
            base = (unsigned long long)__shfl((long long)base, hbase, 64);
            if ((long long)base + kept > a.sub_cap) ovf = true;
            out_off = a.pool_base + (int64_t)sp * a.sub_cap + (int64_t)base;
        }
    } else if (run) {
        // (keeps the shuffle above convergent: nothing to do)
    }
    unsigned long long hsum = 0ULL;
    {
        int at = 0;
        const bool wr = fin && !ovf;
        for (int e0 = 1; __any(wr && e0 < head); e0 += G) {
            const int e = e0 + hl;
            const bool keep = wr && e < head && !((ACC[e >> 5] >> (e & 31)) & 1u);
            const unsigned int km = hb(__ballot(keep));
            if (keep) {
                const int p = at + (int)__popc(km & below);
                pool[out_off + p] = Q[e];
                hsum += mix64(((unsigned long long)(unsigned int)Q[e] << 32) | (unsigned int)p);
            }
            at += (int)__popc(km);
        }
    }
    if (run) hash = 2ULL + (unsigned long long)half_sum((int)(hsum & 0x7fffffffULL)) + (unsigned long long)half_sum((int)((hsum >> 31) & 0x7fffffffULL)) * 0x9E3779B1ULL +
                    (unsigned long long)kept * 0x85EBCA6BULL;
    else { (void)half_sum(0); (void)half_sum(0); }
    if (ovf && hl == 0) atomicOr(&a.st->overflow, tail + 68 > QCAP ? 1 : 2);
    if (hl == 0 && tail > a.st->max_tail) atomicMax(&a.st->max_tail, tail);  // (a statistic: rarely more than a few updates per pass)
    if (hl == 0 && valid_c) {
        const bool same = ei.ns == nsz && ei.cnt == cnt && ei.len == out_len && ei.hash == hash;
        NodeE *w = a.wr + i;
        w->ns = nsz; w->cnt = cnt; w->len = out_len;
        *reinterpret_cast<ulonglong2 *>(&w->off) = make_ulonglong2((unsigned long long)out_off, hash);
        a.cnt_slot_wr[s] = cnt;
        if (!same) a.st->changed[pass & 7] = 1;
#ifdef SVX_MEASURE_PREFIX
        if (!same) { atomicMin(&a.st->minchg[pass & 7], i); atomicAdd(&a.st->nchg[pass & 7], 1); }
#endif
    }
}
// ---- the narrow shape, round 6: one DPP row (16 lanes) per centre, the visited set as a HASH SET in LDS ---------------------------------
// The counters of round 5's kernel (profiles/r6_svx_eval_counters.json) put it at 65 % vector issue with waves parked on memory the rest of
// the time: ~1200 vector and ~600 scalar instructions per wavefront, three quarters of them the visited-set bookkeeping -- a linear scan of the
// queue per list chunk (tail / 4 LDS reads and 4 compares each) and a serial first-occurrence loop of ballots and shuffles.  Here: a byte-wide
// open-addressing table beside the queue (the node's position in the queue, 0xff = empty; a look-up is ~1.1 probes), first occurrences inside a
// chunk by 15 row_shr compares without a loop, row sums by row_ror.  Same queue order, same visited sets, same outputs as eval_kernel<16, 256>.
template <int D> __device__ __forceinline__ unsigned int row_shr_u(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + D, 0xf, 0xf, true);  // (lanes below D of the row read 0)
}
template <int D> __device__ __forceinline__ bool dup_below(unsigned int key) {
    bool d = row_shr_u<D>(key) == key;
    if constexpr (D < 15) d |= dup_below<D + 1>(key);
    return d;
}
template <int D> __device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + D, 0xf, 0xf, true); }
__device__ __forceinline__ int row_sum16(int v) {
    v += row_ror_i<8>(v); v += row_ror_i<4>(v); v += row_ror_i<2>(v); v += row_ror_i<1>(v);
    return v;  // (every lane of the row holds the row's total)
}
constexpr int Q16 = 256, H16 = 256;
__global__ __launch_bounds__(EVAL_WAVES * 64, SVX_EVAL_WPE) void eval16_kernel(FuseArgs a, double lambda, long long budget_total, int pass) {
    constexpr int G = 16, CPW = 4;
    __shared__ int32_t q_all[EVAL_WAVES][CPW][Q16];
    __shared__ unsigned int h_all[EVAL_WAVES][CPW][H16 / 4];
    __shared__ unsigned int acc_all[EVAL_WAVES][CPW][Q16 / 32];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id(), h = lane >> 4, hl = lane & 15, hbase = h * G;
#ifndef SVX_NO_XCD_MAP
    // Workgroups are handed to the 8 XCDs round robin, and each XCD has an L2 of its own: with the plain mapping the sixteen centres of
    // a workgroup -- neighbours in space wherever the cloud's order is coherent (tiles as PCL's voxel filter or a scanner leave them) --
    // have their neighbourhood's records pulled into EIGHT caches.  Here an XCD works through ONE contiguous eighth of the centres.
    // (block 8 q + x is the q-th block of XCD x; XCD x owns nb / 8 blocks, one more when x < nb % 8: a bijection onto [0, nb))
    const unsigned int nb = gridDim.x, x8 = blockIdx.x & 7u, rem = nb & 7u;
    const unsigned int blk = x8 * (nb >> 3) + (x8 < rem ? x8 : rem) + (blockIdx.x >> 3);
#else
    const unsigned int blk = blockIdx.x;
#endif
    const int64_t s0 = ((int64_t)blk * EVAL_WAVES + wave) * CPW;
    if (s0 >= a.nreps) return;  // (whole wave)
    if (a.st->done) return;     // (converged earlier in this batch of passes)
    const int64_t s = s0 + h;
    const bool valid_c = s < a.nreps;
    int32_t *Q = q_all[wave][h];
    unsigned int *HW = h_all[wave][h];
    const unsigned char *HB = reinterpret_cast<const unsigned char *>(HW);
    unsigned int *ACC = acc_all[wave][h];
    const int32_t i = a.reps[valid_c ? s : s0];
    const unsigned int ui = (unsigned int)i;
    const NodeE *__restrict__ E = a.rd;
    const unsigned int *__restrict__ ABS = a.abs_rd;
    int32_t *__restrict__ pool = const_cast<int32_t *>(a.lists);
    const unsigned int below = (1u << hl) - 1u;
    auto hb = [&](unsigned long long m) { return (unsigned int)(m >> hbase) & 0xffffu; };
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const NodeE ei = load_e(E + i);
    const NodeS si = load_s(a.S + i);
    const unsigned int absi = a.abs_rd[i];
    const bool dead = absi != NONE && absi < ui;  // absorbed before its turn: adjacents[i] is empty by then (:121)
    const int64_t off0 = si.off;
    const int len0 = si.len;
    long long budget = a.before ? budget_total - (long long)a.before[valid_c ? s : s0] : 0x7fffffffffffLL;
    const bool run = valid_c && !dead && len0 > 0 && budget > 0;
    int nsz = si.sz, cnt = 0;
    int head = 0, tail = 0;
    bool ovf = false;
    const float pi_[3] = {si.x, si.y, si.z};
    const double ni_[3] = {si.nx, si.ny, si.nz};
    auto slot_of = [](unsigned int r) { return (r * 0x9E3779B1u) >> 24; };
    // the node at queue position `pos` enters the table (lanes of a chunk insert side by side: a byte is claimed by a compare-and-swap of its word)
    auto insert = [&](bool on, unsigned int r, int pos) {
        unsigned int slot = slot_of(r);
        while (on) {
            const unsigned int w = HW[slot >> 2], sh = (slot & 3u) * 8u;
            if (((w >> sh) & 0xffu) == 0xffu) {
                if (atomicCAS(&HW[slot >> 2], w, (w & ~(0xffu << sh)) | ((unsigned int)pos << sh)) == w) on = false;
            } else slot = (slot + 1u) & (unsigned int)(H16 - 1);
        }
    };
    // one chunk of resolved entries joins the queue: the ones not yet visited, in list order (:126-133 / :151-157)
    auto commit = [&](bool go, bool have, unsigned int r) {
        bool fresh = false;
        if (have) {  // visited already?  (the table holds every node of the queue; at most 252 of its 256 bytes are taken)
            fresh = true;
            for (unsigned int slot = slot_of(r);; slot = (slot + 1u) & (unsigned int)(H16 - 1)) {
                const unsigned int p = HB[slot];
                if (p == 0xffu) break;
                if ((unsigned int)Q[p] == r) { fresh = false; break; }
            }
        }
        // ... or by a lower lane of this chunk (the first occurrence wins)
        const bool first = fresh && !dup_below<1>(fresh ? r + 1u : 0u);
        const unsigned int news = hb(__ballot(first));
        const int nn = (int)__popc(news);
        bool put = false;
        int pos = 0;
        if (go) {
            if (tail + nn + 4 > Q16) ovf = true;
            else {
                put = first;
                pos = tail + (int)__popc(news & below);
                if (put) Q[pos] = (int32_t)r;
                tail += nn;
            }
        }
        insert(put, r, pos);
        wsync();
    };
    // appends Find(list entries) that are not yet visited; `act`: this row has a list to append.  TWO chunks of 16 entries per trip:
    // the kernel waits on its dependent gathers (entry -> claim -> claim ...), and the two chunks' chains do not depend on each other --
    // only their order in the queue does, which the commits keep -- so their loads are in flight together.
    auto append_list = [&](bool act, int64_t off, int len) {
        for (int c0 = 0; __any(act && !ovf && c0 < len); c0 += 2 * G) {
            const bool go = act && !ovf && c0 < len;
            const bool haveA = go && c0 + hl < len, haveB = go && c0 + G + hl < len;
            unsigned int rA = NONE, rB = NONE;
            if (haveA) rA = (unsigned int)pool[off + c0 + hl];  // (round-start roots: rootlists_kernel)
            if (haveB) rB = (unsigned int)pool[off + c0 + G + hl];
            // Find as centre i sees it (set.Find, :127/:152): every absorption by a centre that ran BEFORE i and was alive at its own
            // turn; honoured claims lead to ever higher centres, so the walk ends
            unsigned int cA = haveA ? ABS[rA] : NONE, cB = haveB ? ABS[rB] : NONE;
            bool wA = cA != NONE && cA < ui, wB = cB != NONE && cB < ui;
            while (wA || wB) {
                const unsigned int ccA = wA ? ABS[cA] : NONE, ccB = wB ? ABS[cB] : NONE;
                if (wA) {
                    if (ccA != NONE && ccA < cA) wA = false;  // (cA was absorbed before its own turn: its claim does not count)
                    else { rA = cA; cA = ccA; wA = cA != NONE && cA < ui; }
                }
                if (wB) {
                    if (ccB != NONE && ccB < cB) wB = false;
                    else { rB = cB; cB = ccB; wB = cB != NONE && cB < ui; }
                }
            }
            commit(go, haveA, rA);
            if (__any(go && !ovf && c0 + G < len)) commit(go && !ovf && c0 + G < len, haveB && !ovf, rB);
        }
    };

    if (run) {
        if (hl == 0) Q[0] = i;  // visited[i] = true; queue[front++] = i (:123-125)
        for (int e = hl; e < Q16 / 32; e += G) ACC[e] = 0u;
        for (int e = hl; e < H16 / 4; e += G) HW[e] = 0xffffffffu;
        head = tail = 1;
    }
    wsync();
    insert(run && hl == 0, ui, 0);
    wsync();
    append_list(run, off0, len0);
    bool stop = false;
    while (__any(run && !stop && !ovf && head < tail)) {  // :137-163, 16 entries of the queue at a time
        const bool go = run && !stop && !ovf && head < tail;
        const int m = go ? (tail - head < G ? tail - head : G) : 0;
        const bool mine = hl < m;
        const int32_t j = mine ? Q[head + hl] : i;
        // sizes[j] and adjacents[j] as centre i finds them (:142, :151): j ran before i -- grew, kept a list of its own -- iff j < i
        int sj = 0, jlen = 0;
        int64_t joff = 0;
        bool acc = false;
        if (mine) {
            const NodeS sn = load_s(a.S + j);
            sj = sn.sz; joff = sn.off; jlen = sn.len;
            if ((unsigned int)j < ui) {
                const NodeE en = load_e(E + j);  // (the sector its claim was read from a moment ago)
                sj = en.ns; joff = en.off; jlen = en.len;
            }
            const float pj[3] = {sn.x, sn.y, sn.z};
            const double nj[3] = {sn.nx, sn.ny, sn.nz};
            // (measured and not kept, round 6: the test decided in float32 wherever it is clear of the threshold by more than float32 can be
            //  off, the double metric -- a double sqrt and division -- only in the band around it: 224.5 against 216.4 ms per 10 M points;
            //  the kernel waits on its gathers, and the seven registers the second path costs weigh more than the ~65 double-rate
            //  instructions it saves)
            const double loss = (double)sj * sv_metric_vals(pi_, ni_, pj, nj, a.resolution);  // :142 `sizes[j] * metric(points[i], points[j])`
            acc = lambda - loss > 0.0;                                                        // :143-144
        }
        unsigned int accm = hb(__ballot(acc));
        int done = m;  // entries of this chunk that the reference's loop reaches
        if (go && (long long)__popc(accm) >= budget) {  // `if (--number_of_supervoxels == n_supervoxels) break;` (:160) inside this chunk
            unsigned int t = accm;
            for (long long b = 1; b < budget; ++b) t &= t - 1u;
            const int last = __ffs((int)t) - 1;  // lane of the absorption that reaches K
            done = last + 1;
            accm &= (1u << (last + 1)) - 1u;
            stop = true;
        }
        const bool take = mine && ((accm >> hl) & 1u);
        if (take) {
            atomicMin(&a.abs_wr[j], ui);  // set.Link(j, i) (:145); of several centres that claim j in an estimate the lowest counts
            atomicOr(&ACC[(head + hl) >> 5], 1u << ((head + hl) & 31));
        }
        nsz += row_sum16(take ? sj : 0);   // sizes[i] += sizes[j] (:147)
        const int na = (int)__popc(accm);
        cnt += na;
        budget -= na;
        // the absorbed nodes' lists join the queue, in the queue's order (:149-157)
        unsigned int mm = accm;
        while (__any(mm != 0u && !ovf)) {
            const bool ex = mm != 0u && !ovf;
            const int l = ex ? __ffs((int)mm) - 1 : 0;
            if (ex) mm &= mm - 1u;
            const long long o = __shfl((long long)joff, hbase + l, 64);
            const int ln = __shfl(jlen, hbase + l, 64);
            append_list(ex, (int64_t)o, ln);
        }
        if (go) head += done;
    }
    // the centre's list after its turn (:164 `adjacents[i].swap(adjacent)`): the entries it looked at and did not absorb, in order
    int64_t out_off = dead ? 0 : off0;
    int out_len = dead ? 0 : len0;  // (a centre that did not run keeps its list; an absorbed one's is cleared, :158)
    unsigned long long hash = dead ? 1ULL : ~0ULL;
    wsync();
    const bool fin = run && !ovf;
    int kept = 0;
    for (int e0 = 1; __any(fin && e0 < head); e0 += G) {
        const int e = e0 + hl;
        kept += (int)__popc(hb(__ballot(fin && e < head && !((ACC[e >> 5] >> (e & 31)) & 1u))));
    }
    if (fin) {
        out_len = kept;
        out_off = 0;
    }
    {
        const bool alloc = fin && kept > 0;
        const int sp = (int)(s & (SUBPOOLS - 1));
        unsigned long long base = 0ULL;
        if (alloc && hl == 0) base = atomicAdd(&a.st->sub[a.pool_sel][sp], (unsigned long long)kept);
        base = (unsigned long long)__shfl((long long)base, hbase, 64);
        if (alloc) {
            if ((long long)base + kept > a.sub_cap) ovf = true;
            out_off = a.pool_base + (int64_t)sp * a.sub_cap + (int64_t)base;
        }
    }
    unsigned long long hsum = 0ULL;
    {
        int at = 0;
        const bool wr = fin && !ovf;
        for (int e0 = 1; __any(wr && e0 < head); e0 += G) {
            const int e = e0 + hl;
            const bool keep = wr && e < head && !((ACC[e >> 5] >> (e & 31)) & 1u);
            const unsigned int km = hb(__ballot(keep));
            if (keep) {
                const int p = at + (int)__popc(km & below);
                pool[out_off + p] = Q[e];
                hsum += mix64(((unsigned long long)(unsigned int)Q[e] << 32) | (unsigned int)p);
            }
            at += (int)__popc(km);
        }
    }
    {
        const int h0 = row_sum16((int)(hsum & 0x7fffffffULL)), h1 = row_sum16((int)((hsum >> 31) & 0x7fffffffULL));
        if (run) hash = 2ULL + (unsigned long long)h0 + (unsigned long long)h1 * 0x9E3779B1ULL + (unsigned long long)kept * 0x85EBCA6BULL;
    }
    if (ovf && hl == 0) atomicOr(&a.st->overflow, tail + 68 > Q16 ? 1 : 2);
    if (hl == 0 && tail > a.st->max_tail) atomicMax(&a.st->max_tail, tail);  // (a statistic: rarely more than a few updates per pass)
    if (hl == 0 && valid_c) {
        const bool same = ei.ns == nsz && ei.cnt == cnt && ei.len == out_len && ei.hash == hash;
        NodeE *w = a.wr + i;
        w->ns = nsz; w->cnt = cnt; w->len = out_len;
        *reinterpret_cast<ulonglong2 *>(&w->off) = make_ulonglong2((unsigned long long)out_off, hash);
        a.cnt_slot_wr[s] = cnt;
        if (!same) a.st->changed[pass & 7] = 1;
#ifdef SVX_MEASURE_PREFIX
        if (!same) { atomicMin(&a.st->minchg[pass & 7], i); atomicAdd(&a.st->nchg[pass & 7], 1); }
#endif
    }
}
// (the claims: compared after the pass, when all of them are in; and the pass's absorptions in all)
__global__ void abs_changed_kernel(FuseArgs a, int pass, int budget_on, long long budget_total) {
    State *st = a.st;
    if (st->done) return;
    bool ch = false;
    long long tot = 0;
    SVX_FOR(s, a.nreps) {
        const int32_t i = a.reps[s];
        const unsigned int c0 = a.abs_rd[i], c1 = a.abs_wr[i];
        ch = ch || c0 != c1;
#ifdef SVX_MEASURE_PREFIX
        if (c0 != c1) {  // the claimant(s) whose claim on i came or went
            atomicMin(&st->minchg[pass & 7], (int)(c0 < c1 ? c0 : c1)); atomicAdd(&st->nchg[pass & 7], 1);
        }
#endif
        a.abs_rd[i] = NONE;  // (this estimate is the one the NEXT pass writes: its claims start empty)
        tot += a.cnt_slot_wr[s];
    }
    if (__ballot(ch) != 0ULL && lane_id() == 0) st->changed[pass & 7] = 1;
    int lo = wave_sum((int)(tot & 0xffff)), hi = wave_sum((int)(tot >> 16));  // (a lane's share is far below 2^31)
    if (lane_id() == 0 && (lo || hi)) atomicAdd((unsigned long long *)&st->total[pass & 7], (unsigned long long)(((long long)hi << 16) + lo));
    // The last workgroup through closes the pass: did it change nothing (and did it run with the budget, if the budget binds)?  Then
    // the passes the host has queued behind it return at once -- the host looks at the state only every few passes (a look costs a
    // fifth of a pass) -- else it prepares the next pass: its pool's bump pointers, its flag and its total.
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&st->arrive, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    {   // (statistics of the pass's list pool, for F4L_SV_EXACT_DEBUG and the sizing of the pools)
        unsigned long long mx = 0ULL, sm = 0ULL;
        for (int t = (int)threadIdx.x; t < SUBPOOLS; t += (int)blockDim.x) { const unsigned long long v = st->sub[pass & 1][t]; mx = v > mx ? v : mx; sm += v; }
        atomicMax(&st->pool_max, mx);
        __shared__ unsigned long long s_sum;
        if (threadIdx.x == 0) s_sum = 0ULL;
        __syncthreads();
        atomicAdd(&s_sum, sm);
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&st->pool_sum, s_sum);
    }
    const int changed = atomicAdd(&st->changed[pass & 7], 0);
    const long long total = (long long)atomicAdd((unsigned long long *)&st->total[pass & 7], 0ULL);
    const bool conv = changed == 0 && (budget_on || total < budget_total);
    const int next = pass + 1;
    if (!conv)
        for (int t = (int)threadIdx.x; t < SUBPOOLS; t += (int)blockDim.x) st->sub[next & 1][t] = 0ULL;
    if (threadIdx.x == 0) {
        st->arrive = 0u;
        if (conv) { st->done = 1; st->done_it = next; }
        else {
            st->changed[next & 7] = 0; st->total[next & 7] = 0;
#ifdef SVX_MEASURE_PREFIX
            st->minchg[next & 7] = 0x7fffffff; st->nchg[next & 7] = 0;
#endif
        }
    }
}
// the converged estimate becomes the state: survivors keep size and list, everybody follows its absorber
__global__ void commit_kernel(FuseArgs a, int32_t *keep_flag) {
    SVX_FOR(s, a.nreps) {
        const int32_t i = a.reps[s];
        const NodeE e = load_e(a.rd + i);
        const bool survives = a.abs_rd[i] == NONE;
        keep_flag[s] = survives ? 1 : 0;
        NodeS *si = a.S + i;
        if (survives) { si->sz = e.ns; si->off = e.off; si->len = e.len; }
        else si->len = 0;
    }
}
__global__ void reroot_kernel(int32_t *root, const unsigned int *__restrict__ abs, int64_t n) {
    SVX_FOR(x, n) {
        unsigned int r = (unsigned int)root[x];
        for (;;) {  // (a converged estimate: every claim real, chains ascend in time)
            const unsigned int c = abs[r];
            if (c == NONE) break;
            r = c;
        }
        root[x] = (int32_t)r;
    }
}
// the survivors' lists for the next round: every entry replaced by its representative (Find at the start of the round: what every
// evaluation of every pass would otherwise look up per entry).  16 lanes per list; in place (a list has one owner).
__global__ __launch_bounds__(256) void rootlists_kernel(const NodeS *__restrict__ S, int32_t *__restrict__ lists, const int32_t *__restrict__ root,
                                                        const int32_t *__restrict__ reps, int nreps) {
    const int sub = (int)(threadIdx.x & 15);
#ifndef SVX_NO_XCD_MAP  // (an XCD: one contiguous eighth of the representatives -- see SVX_FOR_XCD)
    const bool x = xcd_grid();
    const int64_t per = x ? (((int64_t)nreps + 7) >> 3) : (int64_t)nreps, s_lo = x ? (int64_t)(blockIdx.x & 7u) * per : 0,
                  s_hi = s_lo + per < nreps ? s_lo + per : (int64_t)nreps, s_st = ((int64_t)(x ? gridDim.x >> 3 : gridDim.x) * blockDim.x) >> 4;
    for (int64_t s = s_lo + (((int64_t)(x ? blockIdx.x >> 3 : blockIdx.x) * blockDim.x + threadIdx.x) >> 4); s < s_hi; s += s_st) {
#else
    for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; s < nreps; s += ((int64_t)gridDim.x * blockDim.x) >> 4) {
#endif
        const NodeS *si = S + reps[s];
        const int64_t off = si->off;
        const int len = si->len;
        for (int e = sub; e < len; e += 16) lists[off + e] = root[lists[off + e]];
    }
}
__global__ void compact_kernel(const int32_t *reps, const int32_t *keep_flag, const int32_t *keep_pos, int nreps, int32_t *reps_out) {
    SVX_FOR(s, nreps) if (keep_flag[s]) reps_out[keep_pos[s]] = reps[s];
}

// ---- exchange -----------------------------------------------------------------------------------------------------------
struct XchArgs {
    const float *xyz;
    const double *nrm;
    const int32_t *knn;
    int64_t n;
    int k;
    double resolution;
    State *st;
    // (measured twice and not kept, round 6: position, label and the two estimates as ONE 16-byte record per node -- a neighbour costs
    //  one gather instead of two dependent ones, but 4 nodes share a 64-byte line instead of 16: xch_eval_kernel 37.0 against 28.6 ms per
    //  10 M points under the plain grid stride (HBM bound), and 27.8 against 24.0 ms with an XCD's workgroups side by side (L2 hits))
    int32_t *lab;                 // [n] label = representative point (the generation's starting state)
    double *dis;                  // [n] metric to it
    int32_t *pos;                 // [n] position in the current generation, POS_INF
    unsigned long long *key;      // [n] (pusher position, slot) that first pushes the node into the next generation, KEY_INF
    const int32_t *Q;             // the generation's entries
    int32_t *out_rd, *out_wr;     // [n] an entry's label after its turn (estimate read / written)
    double *dis_wr;
    unsigned char *ch_rd, *ch_wr; // [n] `change` of :214-226
};
#pragma clang fp contract(off)
__global__ void xch_init_kernel(XchArgs a, const int32_t *__restrict__ root) {
    SVX_FOR_XCD(i, a.n) {
        const int32_t l = root[i];
        a.lab[i] = l;
        a.dis[i] = sv_metric(a.xyz, a.nrm, i, (int64_t)l, a.resolution);  // :186-189
        a.pos[i] = POS_INF;
        a.key[i] = KEY_INF;
    }
}
// the scan of :194-207: a point enters the queue at the first event that touches it
__global__ __launch_bounds__(256) void xch_first_keys_kernel(XchArgs a, const int32_t *__restrict__ root) {
    const int sub = (int)(threadIdx.x & 15);
    SVX_FOR_XCD_ROWS(i, a.n) {  // (16 lanes per point; the keys are minima: any order -- the point's own key: one atomic per row)
        const int32_t li = root[i];
        unsigned long long mine = KEY_INF;
        for (int j = sub; j < a.k; j += 16) {
            const int32_t q = a.knn[i * a.k + j];
            if (li != root[q]) {
                const unsigned long long e = ((unsigned long long)i * 64ULL + (unsigned long long)j) * 2ULL;
                mine = e < mine ? e : mine;
                atomicMin(&a.key[q], e + 1ULL);
            }
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
            const unsigned long long o = __shfl_xor(mine, x, 64);
            mine = o < mine ? o : mine;
        }
        if (sub == 0 && mine != KEY_INF) atomicMin(&a.key[i], mine);
    }
}
__global__ void xch_pairs_kernel(const unsigned long long *__restrict__ key, int64_t n, unsigned long long *__restrict__ k_out, int32_t *__restrict__ v_out) {
    SVX_FOR(i, n) { k_out[i] = key[i]; v_out[i] = (int32_t)i; }
}
// after the sort: the entries with a finite key are the generation; their positions; the keys are cleared for the next one
__global__ void xch_generation_kernel(XchArgs a, const unsigned long long *__restrict__ sorted_keys, const int32_t *__restrict__ sorted_nodes) {
    SVX_FOR(t, a.n) {
        const bool in = sorted_keys[t] != KEY_INF;
        if (in) {
            const int32_t i = sorted_nodes[t];
            a.pos[i] = (int32_t)t;
            a.out_rd[i] = a.lab[i];
            a.ch_rd[i] = 0;
            a.key[i] = KEY_INF;
            if (t + 1 == a.n || sorted_keys[t + 1] == KEY_INF) a.st->m = (int)(t + 1);
        } else if (t == 0) a.st->m = 0;
    }
}
// One entry's turn (:214-226) with the labels its earlier entries leave.  The reference walks the neighbours in order and takes
// a label whose representative is STRICTLY closer than the best so far: the result is the label of smallest metric below the
// point's own, the first neighbour slot among equal minima -- a (metric, slot) minimum, taken here by 16 lanes per entry (a
// coalesced 4 k-byte row per entry, four entries per wavefront) and reduced across them.
constexpr int XCH_PER_WG = 64;  // entries of a generation per workgroup of xch_eval_kernel
__global__ __launch_bounds__(256) void xch_eval_kernel(XchArgs a, int m, int pass) {
    // (the host looks at the flags every second pass: a pass queued behind the one that changed nothing returns at once)
    if (pass > 0 && a.st->changed[(pass - 1) & 7] == 0) return;
    const int lane = lane_id(), sub = lane & 15;
    bool differs = false;
    // (a workgroup takes 64 consecutive entries of the generation, and an XCD -- whose L2 is its own -- one contiguous eighth of it with
    //  its resident workgroups side by side: the generation is in (position of the pusher, slot) order, i.e. neighbours in the cloud's
    //  order sit side by side; see eval16_kernel.  Before round 6's last change a workgroup walked a 2048th of the generation:
    //  an XCD's 256 resident workgroups in 256 different places, 167 GB fetched per 10 M points at 5.4 TB/s)
#ifndef SVX_NO_XCD_MAP
    const unsigned int nb = gridDim.x, x8 = blockIdx.x & 7u, rem = nb & 7u;
    const int64_t blk = (int64_t)(x8 * (nb >> 3) + (x8 < rem ? x8 : rem) + (blockIdx.x >> 3));
    const int64_t per_wg = XCH_PER_WG;
#else
    const unsigned int nb = gridDim.x;
    const int64_t blk = blockIdx.x;
    const int64_t per_wg = ((((int64_t)m + nb - 1) / nb + 15) / 16) * 16;
#endif
    const int64_t w_lo = blk * per_wg, w_hi = w_lo + per_wg < m ? w_lo + per_wg : (int64_t)m;
    for (int64_t t0 = w_lo + (int64_t)(threadIdx.x >> 6) * 4; t0 < w_hi; t0 += (int64_t)(blockDim.x >> 6) * 4) {  // (whole waves iterate together)
        const int64_t t = t0 + (lane >> 4);
        const bool valid = t < m;
        const int32_t i = a.Q[valid ? t : 0];
        const int32_t la0 = a.lab[i];
        const double d00 = a.dis[i];
        double bd = d00;
        int bj = 0x7fffffff;
        int32_t bl = la0;
        for (int j = sub; j < a.k; j += 16) {
            const int32_t q = a.knn[(int64_t)i * a.k + j];
            if (q == i) continue;
            const int32_t b = a.pos[q] < (int32_t)t ? a.out_rd[q] : a.lab[q];
            if (b == la0) continue;  // (its own label: never strictly closer than itself)
            const double d = sv_metric(a.xyz, a.nrm, (int64_t)i, (int64_t)b, a.resolution);
            if (d < bd) { bd = d; bj = j; bl = b; }  // (slots ascend within a lane: the first of equal minima stays)
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
            const double od = __shfl_xor(bd, x, 64);
            const int oj = __shfl_xor(bj, x, 64);
            const int32_t ol = __shfl_xor(bl, x, 64);
            if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; bl = ol; }
        }
        const bool ch = bj != 0x7fffffff;  // (some label strictly closer than the point's own)
        if (valid && sub == 0) {
            differs = differs || a.out_rd[i] != bl || (a.ch_rd[i] != 0) != ch;
            a.out_wr[i] = bl;
            a.dis_wr[i] = bd;
            a.ch_wr[i] = ch ? 1 : 0;
        }
    }
    if (__ballot(differs) != 0ULL && lane_id() == 0) a.st->changed[pass & 7] = 1;
}
// the pushes of the generation (:228-236) and its labels
__global__ void xch_push_kernel(XchArgs a, int m) {
    SVX_FOR_XCD(t, m) {
        const int32_t i = a.Q[t];
        const int32_t li = a.out_rd[i];
        if (!a.ch_rd[i]) continue;
        for (int j = 0; j < a.k; ++j) {
            const int32_t q = a.knn[(int64_t)i * a.k + j];
            if (q == i) continue;
            const int32_t pq = a.pos[q];
            const int32_t b = pq < (int32_t)t ? a.out_rd[q] : a.lab[q];
            if (li != b && !(pq != POS_INF && pq > (int32_t)t)) atomicMin(&a.key[q], (unsigned long long)t * 64ULL + (unsigned long long)j);
        }
    }
}
__global__ void xch_commit_kernel(XchArgs a, int m, const double *__restrict__ dis_rd) {
    SVX_FOR(t, m) {
        const int32_t i = a.Q[t];
        a.lab[i] = a.out_rd[i];
        a.dis[i] = dis_rd[i];
        a.pos[i] = POS_INF;
    }
}
__global__ void rank_kernel(const int32_t *__restrict__ reps, int nreps, int32_t *__restrict__ rank) {
    SVX_FOR(s, nreps) rank[reps[s]] = (int32_t)s;
}
__global__ void relabel_kernel(const int32_t *__restrict__ lab, const int32_t *__restrict__ rank, int64_t n, int32_t *__restrict__ labels_out) {
    SVX_FOR(i, n) labels_out[i] = rank[lab[i]];  // :241-247
}

static inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }
struct Ws {
    State *st;
    int32_t *lists;  // [n k] the neighbour table (a copy: one address space with the pools) + 3 pools
    int64_t pool_off[3], pool_cap;
    NodeS *S;
    NodeE *E[2];
    unsigned int *abs[2];
    int32_t *root, *reps_a, *reps_b, *keep_flag, *keep_pos, *cnt_slot[2];
    int64_t *before;
    double *dis, *dis2[2], *median;
    void *sel;
    unsigned long long *key, *key_s_in, *key_s_out;
    int32_t *node_s_in, *node_s_out, *pos, *lab, *out[2], *rank;
    unsigned char *ch[2];
    void *prim;
    size_t prim_bytes, total;
};
static int layout(int64_t n, int k, Ws &w, unsigned char *base) {
    size_t sort_b = 0, scan_b = 0, scan64_b = 0;
    unsigned long long *k0 = nullptr;
    int32_t *i0 = nullptr;
    int64_t *l0 = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, sort_b, k0, k0, i0, i0, (size_t)n, 0, 40, 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::exclusive_scan(nullptr, scan_b, i0, i0, 0, (size_t)n, rocprim::plus<int32_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    if (rocprim::exclusive_scan(nullptr, scan64_b, i0, l0, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), 0, false) != hipSuccess) return F4L_EHIP;
    size_t prim = sort_b > scan_b ? sort_b : scan_b;
    prim = prim > scan64_b ? prim : scan64_b;
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return base ? base + at : (unsigned char *)nullptr; };
    const size_t nk = (size_t)n * (size_t)k, N = (size_t)n;
    w.st = (State *)carve(sizeof(State));
    // the pools: a centre's list is inherited by ONE absorber in a consistent estimate, so the lists of a pass sum to at most the
    // lists before it (n k at the start); an inconsistent estimate may count some twice: room for 2 n k, in SUBPOOLS slices
    w.pool_cap = (int64_t)(((2 * nk + 2 * N) / SUBPOOLS + 64) * SUBPOOLS);  // (+ what was absorbed and passed through: at most a few per centre)
    w.lists = (int32_t *)carve((nk + 3 * (size_t)w.pool_cap) * 4);
    for (int p = 0; p < 3; ++p) w.pool_off[p] = (int64_t)nk + (int64_t)p * w.pool_cap;
    w.root = (int32_t *)carve(N * 4);
    w.reps_a = (int32_t *)carve(N * 4); w.reps_b = (int32_t *)carve(N * 4);
    w.dis = (double *)carve(N * 8);
    w.dis2[0] = (double *)carve(N * 8); w.dis2[1] = (double *)carve(N * 8);
    w.median = (double *)carve(16);
    w.sel = carve(select_workspace_bytes());
    w.prim = carve(prim);
    w.prim_bytes = prim;
    // what only the fusion needs and what only the exchange needs share one region (the fusion is over when the exchange starts;
    // `root`, the representatives and `dis` live on through both)
    const size_t region = o;
    w.S = (NodeS *)carve(N * sizeof(NodeS));
    w.E[0] = (NodeE *)carve(N * sizeof(NodeE)); w.E[1] = (NodeE *)carve(N * sizeof(NodeE));
    w.abs[0] = (unsigned int *)carve(N * 4); w.abs[1] = (unsigned int *)carve(N * 4);
    w.keep_flag = (int32_t *)carve(N * 4); w.keep_pos = (int32_t *)carve(N * 4);
    w.cnt_slot[0] = (int32_t *)carve(N * 4); w.cnt_slot[1] = (int32_t *)carve(N * 4);
    w.before = (int64_t *)carve(N * 8);
    const size_t fusion_end = o;
    o = region;
    w.key = (unsigned long long *)carve(N * 8); w.key_s_in = (unsigned long long *)carve(N * 8); w.key_s_out = (unsigned long long *)carve(N * 8);
    w.node_s_in = (int32_t *)carve(N * 4); w.node_s_out = (int32_t *)carve(N * 4);
    w.pos = (int32_t *)carve(N * 4); w.lab = (int32_t *)carve(N * 4);
    w.out[0] = (int32_t *)carve(N * 4); w.out[1] = (int32_t *)carve(N * 4);
    w.rank = (int32_t *)carve(N * 4);
    w.ch[0] = carve(N); w.ch[1] = carve(N);
    o = o > fusion_end ? o : fusion_end;
    w.total = o;
    return F4L_OK;
}
}  // namespace svx
}  // namespace f4l

extern "C" size_t f4l_supervoxel_segment_exact_workspace_bytes(int64_t n, int k) {
    if (n <= 0 || k < 1) return 0;
    f4l::svx::Ws w;
    if (f4l::svx::layout(n, k, w, nullptr) != F4L_OK) return 0;
    return w.total;
}

// The reference's segmentation (supervoxel_segmentation.h:65-248) of device arrays, label for label.  SYNCHRONISES `stream` (a
// few times per lambda round and exchange generation: whether the estimate still changes, how many representatives are left).
// F4L_EUNSUPPORTED: a closure or the lists outgrew the device buffers (nothing was written to labels_out): replay on the host.
static int segment_exact_run(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k, double resolution,
                             int32_t *labels_out, int32_t *n_supervoxels_host, int32_t *stats_host, void *workspace, size_t workspace_bytes,
                             void *stream, bool wide, bool &queue_outgrown) {
    using namespace f4l;
    using namespace f4l::svx;
    queue_outgrown = false;
    if (!xyz || !normals || !knn || n <= 0 || k < 1 || !(resolution > 0.0) || !labels_out || !workspace) return F4L_EINVAL;
    if (k > F4L_MAX_K || n > 0x3fffffffLL) return F4L_EUNSUPPORTED;
    Ws w;
    int rc = layout(n, k, w, (unsigned char *)workspace);
    if (rc != F4L_OK) return rc;
    if (workspace_bytes < w.total) return F4L_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(2048), b(256), one(1);
    // (the looks at the state land in PINNED host memory, one buffer per host thread, kept: a copy into pageable memory goes through
    //  the runtime's staging path, and there are a hundred looks per call)
    static thread_local State *hs_pinned = nullptr;
    if (!hs_pinned && hipHostMalloc((void **)&hs_pinned, sizeof(State), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); hs_pinned = nullptr; }
    State hs_pageable;
    State &hs = hs_pinned ? *hs_pinned : hs_pageable;
    auto read_state = [&]() -> int {
        F4L_HIP_CHECK(hipMemcpyAsync(&hs, w.st, offsetof(State, sub), hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipStreamSynchronize(st));
        return F4L_OK;
    };
    F4L_HIP_CHECK(hipMemsetAsync(w.st, 0, sizeof(State), st));
    {
        State init;
        memset(&init, 0, sizeof(init));
        for (int d = 0; d < 3; ++d) { init.bb[d] = 0xffffffffu; init.bb[3 + d] = 0u; }
        F4L_HIP_CHECK(hipMemcpyAsync(w.st, &init, sizeof(unsigned int) * 6, hipMemcpyHostToDevice, st));
    }
    // K (grid_sample.h:48-68): the hash set lives in the first pool, idle until the first pass writes lists
    hipLaunchKernelGGL(svx::bbox_kernel, dim3(256), b, 0, st, xyz, n, w.st);
    {
        const unsigned long long slots = 2ULL * (unsigned long long)n + 1024ULL;
        unsigned long long *set = (unsigned long long *)(w.lists + ((w.pool_off[1] + 1) & ~(int64_t)1));  // (an even entry of a 256-byte aligned array: 8-byte aligned)
        if (((size_t)w.pool_cap - 1) * 4 < (size_t)slots * 8) return F4L_EUNSUPPORTED;  // (k = 1: replay on the host)
        F4L_HIP_CHECK(hipMemsetAsync(set, 0xff, (size_t)slots * 8, st));
        hipLaunchKernelGGL(svx::grid_count_kernel, g, b, 0, st, xyz, n, resolution, w.st, set, slots);
    }
    // lambda0 (:105-113)
    hipLaunchKernelGGL(svx::iota_kernel, g, b, 0, st, w.root, w.reps_a, w.S, xyz, normals, n, k);
    hipLaunchKernelGGL(svx::min_metric_kernel, g, b, 0, st, (const svx::NodeS *)w.S, knn, n, k, resolution, w.dis);
    F4L_LAUNCH_CHECK();
    {
        const int64_t rank = n / 2;
        rc = select_ranks_f64(w.dis, n, 1, 1, &rank, w.median, w.sel, st);
        if (rc != F4L_OK) return rc;
    }
    double lambda0 = 0.0;
    F4L_HIP_CHECK(hipMemcpyAsync(&lambda0, w.median, 8, hipMemcpyDeviceToHost, st));
    F4L_HIP_CHECK(hipMemcpyAsync(w.lists, knn, (size_t)n * k * 4, hipMemcpyDeviceToDevice, st));
    rc = read_state();
    if (rc != F4L_OK) return rc;
    const int K = hs.K;
    // K == n (a resolution below the point spacing: every point in a cell of its own): the reference's `--number == n_supervoxels`
    // (:160) never fires once a first absorption took the count below K -- it then fuses on to ONE supervoxel -- while its check
    // after a centre's turn (:169) stops everything at once when the first centre absorbs nothing (n singletons).  Neither is what
    // a budget of (live - K) = 0 describes: left to the host replay (tests/test_gpu_supervoxel_exact.py).
    if ((int64_t)K >= n) return F4L_EUNSUPPORTED;
    double lambda = lambda0 > DBL_EPSILON ? lambda0 : DBL_EPSILON;

    FuseArgs fa;
    fa.S = w.S; fa.lists = w.lists; fa.n = n; fa.k = k; fa.resolution = resolution; fa.st = w.st;
    fa.sub_cap = w.pool_cap / SUBPOOLS;
    int32_t *reps = w.reps_a, *reps_next = w.reps_b;
    int nreps = (int)n, live = (int)n;
    int committed_pool = -1;  // (the pool the round-start lists live in; -1: the neighbour table)
    const bool old16 = getenv("F4L_SV_EXACT_SCAN") != nullptr;  // (A/B: the narrow shape with round 5's linear visited scan)
    int rounds = 0, passes = 0;
    auto set_est = [&](int rd) { fa.rd = w.E[rd]; fa.wr = w.E[rd ^ 1]; fa.abs_rd = w.abs[rd]; fa.abs_wr = w.abs[rd ^ 1]; fa.cnt_slot_rd = w.cnt_slot[rd]; fa.cnt_slot_wr = w.cnt_slot[rd ^ 1]; };
    for (; rounds < MAX_ROUNDS; lambda *= 2.0, ++rounds) {
        if (nreps <= 1) break;  // :118
        fa.reps = reps; fa.nreps = nreps;
        int rd = 0;
        set_est(0);
        hipLaunchKernelGGL(svx::round_init_kernel, g, b, 0, st, fa);
        const long long budget_total = (long long)live - K;
        // the two pools the estimates' lists alternate between: the ones the round-start lists are not in
        int pe[2], np = 0;
        for (int p = 0; p < 3; ++p) if (p != committed_pool && np < 2) pe[np++] = p;
        long long prev_total = 0;
        bool converged = false;
        int it = 0;
        // passes between two looks at the state: two in the long rounds (a look -- copy, synchronise, restart an empty queue -- costs a
        // fifth of a pass at 1 M centres; per 10 M points 278 ms with one, 260 with two, 282 with three), four in the short late rounds
        // (round 5: and the passes queued behind the one that converges do nothing -- abs_changed_kernel decides -- so a batch costs no wasted pass)
        int batch_len = nreps > 40000 ? 4 : 8;
        if (const char *e = getenv("F4L_SV_EXACT_BATCH")) { const int v = atoi(e); if (v >= 1 && v <= 16) batch_len = v; }  // (measurement)
        while (!converged) {
            bool budget_on = false;
            for (int batch = 0; batch < batch_len; ++batch, ++it, ++passes) {
                if (it >= MAX_ITERS) return F4L_EUNSUPPORTED;
                set_est(rd);
                fa.pool_sel = it & 1; fa.pool_base = w.pool_off[pe[it & 1]];
                // the budget only binds in the round that reaches K: the prefix sum of the absorptions is taken once a pass has
                // absorbed as much as the budget
                fa.before = nullptr;
                budget_on = prev_total >= budget_total;
                if (budget_on) {
                    size_t tb = w.prim_bytes;
                    F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim, tb, fa.cnt_slot_rd, w.before, (int64_t)0, (size_t)nreps, rocprim::plus<int64_t>(), st, false));
                    fa.before = w.before;
                }
                if (wide) hipLaunchKernelGGL((svx::eval_kernel<32, QCAP_WIDE>), dim3((unsigned)((nreps + 2 * EVAL_WAVES - 1) / (2 * EVAL_WAVES))), dim3(EVAL_WAVES * 64), 0, st, fa, lambda, budget_total, it);
                else if (old16) hipLaunchKernelGGL((svx::eval_kernel<16, 256>), dim3((unsigned)((nreps + 4 * EVAL_WAVES - 1) / (4 * EVAL_WAVES))), dim3(EVAL_WAVES * 64), 0, st, fa, lambda, budget_total, it);
                else hipLaunchKernelGGL(svx::eval16_kernel, dim3((unsigned)((nreps + 4 * EVAL_WAVES - 1) / (4 * EVAL_WAVES))), dim3(EVAL_WAVES * 64), 0, st, fa, lambda, budget_total, it);
                hipLaunchKernelGGL(svx::abs_changed_kernel, dim3(nreps > 262144 ? 256 : 64), b, 0, st, fa, it, budget_on ? 1 : 0, budget_total);
                F4L_LAUNCH_CHECK();
                rd ^= 1;
#ifdef SVX_MEASURE_PREFIX
                if (read_state() == F4L_OK)
                    fprintf(stderr, "[sv prefix] round %d pass %d: %d of %d centres: lowest changed %.4f of n, %d changes\n", rounds, it, nreps, (int)n,
                            hs.minchg[it & 7] == 0x7fffffff ? 1.0 : (double)hs.minchg[it & 7] / (double)n, hs.nchg[it & 7]);
#endif
            }
            rc = read_state();
            if (rc != F4L_OK) return rc;
            if (hs.overflow) { queue_outgrown = (hs.overflow & 1) != 0; return F4L_EUNSUPPORTED; }
            if (hs.done) {  // the passes after number done_it did nothing: the estimate is the one pass done_it - 1 wrote
                passes -= it - hs.done_it;
                it = hs.done_it;
                rd = it & 1;
            }
            const long long total = hs.total[(it - 1) & 7];
            prev_total = total;
            // converged: the last pass changed nothing -- and it ran with the budget if the budget binds
            converged = hs.done != 0;
        }
        // commit (the estimate `rd` = the last one written; its lists are in pool pe[(it - 1) & 1])
        set_est(rd);
        hipLaunchKernelGGL(svx::commit_kernel, g, b, 0, st, fa, w.keep_flag);
        hipLaunchKernelGGL(svx::reroot_kernel, g, b, 0, st, w.root, (const unsigned int *)fa.abs_rd, n);
        {
            size_t tb = w.prim_bytes;
            F4L_HIP_CHECK(rocprim::exclusive_scan(w.prim, tb, w.keep_flag, w.keep_pos, 0, (size_t)nreps, rocprim::plus<int32_t>(), st, false));
        }
        hipLaunchKernelGGL(svx::compact_kernel, g, b, 0, st, reps, w.keep_flag, w.keep_pos, nreps, reps_next);
        F4L_LAUNCH_CHECK();
        int32_t lastp = 0, lastf = 0;
        F4L_HIP_CHECK(hipMemcpyAsync(&lastp, w.keep_pos + (nreps - 1), 4, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipMemcpyAsync(&lastf, w.keep_flag + (nreps - 1), 4, hipMemcpyDeviceToHost, st));
        F4L_HIP_CHECK(hipStreamSynchronize(st));
        committed_pool = pe[(it - 1) & 1];
        nreps = lastp + lastf;
        live = nreps;
        { int32_t *t = reps; reps = reps_next; reps_next = t; }
        if (nreps > 1 && nreps != K)  // (another round follows: its lists hold this round's representatives)
            hipLaunchKernelGGL(svx::rootlists_kernel, dim3((unsigned)((((nreps + 15) / 16 < 4096 ? (nreps + 15) / 16 : 4096) + 7) & ~7)), b, 0, st, (const svx::NodeS *)w.S, w.lists,
                               (const int32_t *)w.root, (const int32_t *)reps, nreps);
        if (getenv("F4L_SV_EXACT_DEBUG"))
            fprintf(stderr, "[sv exact] round %d lambda %.6g: %d passes, %d representatives left (K %d); list pools so far: fullest slice %.3f of its capacity, "
                            "largest pass %.3f n k entries\n", rounds, lambda, it, nreps, K, (double)hs.pool_max / (double)fa.sub_cap, (double)hs.pool_sum / ((double)n * k));
        if (nreps == K) { ++rounds; break; }  // :175
    }
    if (rounds >= MAX_ROUNDS && nreps != K && nreps > 1) return F4L_EUNSUPPORTED;

    // ---- the exchange (:186-237)
    XchArgs xa;
    xa.xyz = xyz; xa.nrm = normals; xa.knn = knn; xa.n = n; xa.k = k; xa.resolution = resolution; xa.st = w.st;
    xa.lab = w.lab; xa.dis = w.dis; xa.pos = w.pos; xa.key = w.key;
    hipLaunchKernelGGL(svx::xch_init_kernel, g, b, 0, st, xa, (const int32_t *)w.root);
    hipLaunchKernelGGL(svx::xch_first_keys_kernel, g, b, 0, st, xa, (const int32_t *)w.root);
    F4L_LAUNCH_CHECK();
    int generations = 0, xpasses = 0;
    const bool xch_jacobi = getenv("F4L_SV_EXACT_XCH_JACOBI") != nullptr;
    for (;; ++generations) {
        if (generations >= MAX_GENERATIONS) return F4L_EUNSUPPORTED;
        // the next generation: the nodes with a key, in key order
        hipLaunchKernelGGL(svx::xch_pairs_kernel, g, b, 0, st, (const unsigned long long *)w.key, n, w.key_s_in, w.node_s_in);
        {
            size_t tb = w.prim_bytes;
            F4L_HIP_CHECK(rocprim::radix_sort_pairs(w.prim, tb, w.key_s_in, w.key_s_out, w.node_s_in, w.node_s_out, (size_t)n, 0, 40, st, false));
        }
        xa.Q = w.node_s_out;
        int rd = 0;
        xa.out_rd = w.out[0]; xa.out_wr = w.out[1]; xa.ch_rd = w.ch[0]; xa.ch_wr = w.ch[1]; xa.dis_wr = w.dis2[1];
        hipLaunchKernelGGL(svx::xch_generation_kernel, g, b, 0, st, xa, (const unsigned long long *)w.key_s_out, (const int32_t *)w.node_s_out);
        F4L_LAUNCH_CHECK();
        rc = read_state();
        if (rc != F4L_OK) return rc;
        const int m = hs.m;
        if (m == 0) break;
        bool converged = false;
        int it = 0;
        while (!converged) {
            for (int batch = 0; batch < 2; ++batch, ++it, ++xpasses) {
                if (it >= MAX_ITERS) return F4L_EUNSUPPORTED;
                // (the estimate is updated IN PLACE since round 6: an entry's label is one word, so a reader sees the old or the new one,
                //  and either is an estimate -- the fixed point is unique, recursion on the position in the generation -- while entries whose
                //  earlier neighbours were through before them see their final labels in the same pass: fewer passes.  A pass that changes
                //  nothing wrote nothing, so it read the state every entry agrees with.  F4L_SV_EXACT_XCH_JACOBI=1: two buffers, as before.)
                if (xch_jacobi) { xa.out_rd = w.out[rd]; xa.out_wr = w.out[rd ^ 1]; xa.ch_rd = w.ch[rd]; xa.ch_wr = w.ch[rd ^ 1]; }
                else { xa.out_rd = w.out[0]; xa.out_wr = w.out[0]; xa.ch_rd = w.ch[0]; xa.ch_wr = w.ch[0]; }
                xa.dis_wr = w.dis2[0];  // (written by every pass that runs, read by the commit only)
                F4L_HIP_CHECK(hipMemsetAsync(&w.st->changed[it & 7], 0, 4, st));
#ifndef SVX_NO_XCD_MAP
                hipLaunchKernelGGL(svx::xch_eval_kernel, dim3((unsigned)((m + svx::XCH_PER_WG - 1) / svx::XCH_PER_WG)), b, 0, st, xa, m, it);
#else
                hipLaunchKernelGGL(svx::xch_eval_kernel, g, b, 0, st, xa, m, it);
#endif
                F4L_LAUNCH_CHECK();
                rd ^= 1;
            }
            rc = read_state();
            if (rc != F4L_OK) return rc;
            converged = hs.changed[(it - 1) & 7] == 0;
            if (getenv("F4L_SV_EXACT_DEBUG")) fprintf(stderr, "[sv exact] generation %d (%d entries): passes %d and %d changed something: %d, %d\n", generations, m, it - 2, it - 1, hs.changed[(it - 2) & 7], hs.changed[(it - 1) & 7]);
        }
        if (xch_jacobi) { xa.out_rd = w.out[rd]; xa.ch_rd = w.ch[rd]; }
        else { xa.out_rd = w.out[0]; xa.ch_rd = w.ch[0]; }
        hipLaunchKernelGGL(svx::xch_push_kernel, g, b, 0, st, xa, m);
        hipLaunchKernelGGL(svx::xch_commit_kernel, g, b, 0, st, xa, m, (const double *)w.dis2[0]);
        F4L_LAUNCH_CHECK();
    }
    // ---- relabel (:241-247)
    hipLaunchKernelGGL(svx::rank_kernel, g, b, 0, st, (const int32_t *)reps, nreps, w.rank);
    hipLaunchKernelGGL(svx::relabel_kernel, g, b, 0, st, (const int32_t *)w.lab, (const int32_t *)w.rank, n, labels_out);
    F4L_LAUNCH_CHECK();
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    if (n_supervoxels_host) *n_supervoxels_host = nreps;
    if (stats_host) { stats_host[0] = rounds; stats_host[1] = passes; stats_host[2] = generations; stats_host[3] = xpasses; stats_host[4] = hs.max_tail; }
    if (getenv("F4L_SV_EXACT_DEBUG")) fprintf(stderr, "[sv exact] %d rounds, %d passes, %d generations, %d exchange passes; largest closure %d of %d\n", rounds, passes, generations, xpasses, hs.max_tail, wide ? QCAP_WIDE : 256);
    return F4L_OK;
}

extern "C" int f4l_supervoxel_segment_exact(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k,
                                            double resolution, int32_t *labels_out, int32_t *n_supervoxels_host, int32_t *stats_host,
                                            void *workspace, size_t workspace_bytes, void *stream) {
    bool outgrown = false;
    int rc = segment_exact_run(xyz, normals, knn, n, k, resolution, labels_out, n_supervoxels_host, stats_host, workspace, workspace_bytes,
                               stream, getenv("F4L_SV_EXACT_WIDE") != nullptr, outgrown);
    if (rc == F4L_EUNSUPPORTED && outgrown && !getenv("F4L_SV_EXACT_WIDE")) {  // a closure beyond the narrow queue: once more, wide
        if (getenv("F4L_SV_EXACT_DEBUG")) fprintf(stderr, "[sv exact] a closure beyond 256 nodes: restarting with the wide queue\n");
        rc = segment_exact_run(xyz, normals, knn, n, k, resolution, labels_out, n_supervoxels_host, stats_host, workspace, workspace_bytes, stream,
                               true, outgrown);
    }
    return rc;
}
