// topk.h -- wave-resident exact top-k (k <= 64) for gfx950.
//
// One wavefront owns one query.  The running neighbour list lives in registers, one slot per lane, kept
// sorted ascending by (d2, index): lane L holds the L-th best candidate seen so far.  A batch of 64
// candidates is filtered with one ballot against the current k-th entry; each survivor is inserted with a
// single DPP wave_shr:1 shift of the tail (no LDS, no divergence: every step is wave-uniform).
// Distances are double so that ordering matches the reference's double arithmetic bit for bit.
#pragma once
#include "f4l_device.h"

namespace f4l {

// lane L receives the value of lane L-1 (lane 0 keeps its own): DPP wave_shr:1 (gfx9 encoding 0x138)
__device__ __forceinline__ int dpp_shr1(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ double dpp_shr1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_shr1((int)(b & 0xffffffffLL)), hi = dpp_shr1((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ bool before(double da, int ia, double db, int ib) {
    return da < db || (da == db && ia < ib);
}

struct WaveTopK {
    double d;  // this lane's slot: squared distance
    int i;     // this lane's slot: candidate id
    __device__ __forceinline__ void reset() {
        d = __builtin_inf();
        i = 0x7fffffff;
    }
    // First batch of a query: the list is empty, so instead of inserting the candidates one by one the 64 of them are
    // loaded into the slots and sorted in place by a bitonic network across the lanes (21 compare-exchange stages).
    // The list then holds up to 64 real entries, ascending; entries beyond the k-th are harmless.
    __device__ __forceinline__ void fill_sorted(double cd, int ci) {
        d = cd;
        i = ci;
        const int lane = lane_id();
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const double od = __shfl_xor(d, j, 64);
                const int oi = __shfl_xor(i, j, 64);
                const bool up = (lane & k) == 0;     // this block sorts ascending (k == 64: every lane)
                const bool lower = (lane & j) == 0;  // the lower lane of the pair
                const bool mine_first = before(d, i, od, oi);
                const bool keep_mine = (lower == up) ? mine_first : !mine_first;
                d = keep_mine ? d : od;
                i = keep_mine ? i : oi;
            }
        }
    }
    // Offer one candidate per lane (cd = +inf for idle lanes).  k-1 must be wave-uniform.
    __device__ __forceinline__ void offer(double cd, int ci, int k) {
        double thr = readlane_f64(d, k - 1);
        int thri = __builtin_amdgcn_readlane(i, k - 1);
        unsigned long long mask = __ballot(before(cd, ci, thr, thri));
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const double nd = readlane_f64(cd, j);
            const int ni = __builtin_amdgcn_readlane(ci, j);
            if (!before(nd, ni, thr, thri)) continue;  // the k-th entry tightened meanwhile (uniform)
            const bool mine_first = before(d, i, nd, ni);
            const int pos = __builtin_popcountll(__ballot(mine_first));  // sorted list: a prefix of lanes
            const double pd = dpp_shr1(d);
            const int pi = dpp_shr1(i);
            const int lane = lane_id();
            if (lane > pos) { d = pd; i = pi; }
            else if (lane == pos) { d = nd; i = ni; }
            thr = readlane_f64(d, k - 1);
            thri = __builtin_amdgcn_readlane(i, k - 1);
        }
    }
    __device__ __forceinline__ double kth(int k) const { return readlane_f64(d, k - 1); }
};

// Squared Euclidean distance in double from float coordinates, accumulated x, y, z with separately rounded
// multiply and add (no FMA), as codelibrary/util/metric/squared_euclidean.h:25-36 compiled for x86-64.
__device__ __forceinline__ double dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)  // HIP's __dmul_rn/__dadd_rn are plain operators: keep hipcc from fusing them
    const double dx = (double)ax - (double)bx, dy = (double)ay - (double)by, dz = (double)az - (double)bz;
    double t = dx * dx;
    t = t + dy * dy;
    t = t + dz * dz;
    return t;
}

}  // namespace f4l
