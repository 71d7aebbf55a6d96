// lane_topk.h -- a lane's list of (exact squared distance, index) pairs sorted on registers: the sorting network and the tie
// order of knn_lanes_kernel (csrc/knn.hip), for kernels that keep one query per lane.
#pragma once
#include "f4l_device.h"

namespace f4l {

// compare-exchange on registers, ascending; equal keys keep their places (their order is settled afterwards)
__device__ __forceinline__ void lane_ce(double &ka, int &pa, double &kb, int &pb) {
    const bool sw = kb < ka;
    const double lo = __builtin_fmin(ka, kb), hi = __builtin_fmax(ka, kb);  // (no NaNs here: distances and +inf padding)
    const int plo = sw ? pb : pa, phi = sw ? pa : pb;
    ka = lo; kb = hi; pa = plo; pb = phi;
}

// Bitonic network of 64 with ascending comparators only (per merge size one mirrored stage, then the half-cleaners).  Entries
// CAP..63 would be +inf padding; in an all-ascending network the largest elements at the top never move, so every comparator
// that touches them is a no-op and is left out (and the padding needs no registers).  CAP <= 64.
template <int CAP>
__device__ __forceinline__ void lane_sort_ascending(double (&key)[CAP], int (&pay)[CAP]) {
#pragma unroll
    for (int lm = 1; lm <= 6; ++lm) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int l = i ^ ((1 << lm) - 1);
            if (l > i && l < CAP) lane_ce(key[i], pay[i], key[l < CAP ? l : 0], pay[l < CAP ? l : 0]);
        }
#pragma unroll
        for (int lj = lm - 2; lj >= 0; --lj) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int l = i ^ (1 << lj);
                if (l > i && l < CAP) lane_ce(key[i], pay[i], key[l < CAP ? l : 0], pay[l < CAP ? l : 0]);
            }
        }
    }
}

// Exactly equal keys (gridded or duplicated points): ordered by index.  All lanes of the wave must make the call together.
template <int CAP>
__device__ __forceinline__ void lane_order_ties(double (&key)[CAP], int (&pay)[CAP]) {
    bool anyeq = false;
#pragma unroll
    for (int j = 0; j + 1 < CAP; ++j) anyeq = anyeq | ((key[j] == key[j + 1]) & (pay[j] > pay[j + 1]));  // (padding: equal indices)
    if (__ballot(anyeq) == 0ULL) return;
    for (int pass = 0; pass < CAP; ++pass) {  // rare: odd-even transposition inside groups of equal keys until nothing moves
        bool moved = false;
#pragma unroll
        for (int j = 0; j + 1 < CAP; j += 2) {
            const bool sw = (key[j] == key[j + 1]) & (pay[j] > pay[j + 1]);
            const int lo = sw ? pay[j + 1] : pay[j], hi = sw ? pay[j] : pay[j + 1];
            pay[j] = lo; pay[j + 1] = hi; moved = moved | sw;
        }
#pragma unroll
        for (int j = 1; j + 1 < CAP; j += 2) {
            const bool sw = (key[j] == key[j + 1]) & (pay[j] > pay[j + 1]);
            const int lo = sw ? pay[j + 1] : pay[j], hi = sw ? pay[j] : pay[j + 1];
            pay[j] = lo; pay[j + 1] = hi; moved = moved | sw;
        }
        if (__ballot(moved) == 0ULL) break;
    }
}

}  // namespace f4l
