// select.hip -- the k-th smallest of n doubles by a most-significant-digit radix select.
//
// The medians of the path -- `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754: numpy.median of the
// nearest-neighbour distances) and the starting lambda of the supervoxel fusion (codelibrary/statistics/median.h:27-30:
// nth_element at size / 2) -- need one or two order statistics, not the order of 10 M values: six passes that read the values
// (digits of 11, 11, 11, 11, 10, 10 bits from the top) instead of the eight read-and-scatter passes of a 64-bit radix sort.
// A pass counts, per rank asked for, the digit of every value whose higher digits equal the prefix found so far (histogram in
// LDS, merged into device memory); a one-workgroup kernel then picks the digit whose bucket holds the rank and extends the
// prefix.  After the last pass the prefix IS the value (keys are the doubles' bit patterns made order preserving).
#include "f4l_device.h"
#include "select.h"
#include "../../include/f4l.h"

namespace f4l {
namespace {
constexpr int SEL_BINS = 2048;
struct SelState {
    unsigned long long prefix[SELECT_MAX_RANKS];  // the digits picked so far, right aligned
    unsigned long long k[SELECT_MAX_RANKS];       // rank inside the bucket the prefix names
    unsigned int hist[SELECT_MAX_RANKS][SEL_BINS];
};
__device__ __forceinline__ unsigned long long sel_key(double v) {  // order preserving: negatives flipped, the others offset
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return b ^ ((b >> 63) ? ~0ULL : 0x8000000000000000ULL);
}
__device__ __forceinline__ double sel_value(unsigned long long key) {
    const unsigned long long b = key ^ ((key >> 63) ? 0x8000000000000000ULL : ~0ULL);
    return __longlong_as_double((long long)b);
}
__global__ void select_init_kernel(SelState *s, int n_ranks, long long k0, long long k1) {
    const int t = (int)threadIdx.x;
    for (int r = 0; r < SELECT_MAX_RANKS; ++r)
        for (int b = t; b < SEL_BINS; b += (int)blockDim.x) s->hist[r][b] = 0u;
    if (t == 0) {
        s->prefix[0] = 0ULL; s->prefix[1] = 0ULL;
        s->k[0] = (unsigned long long)k0; s->k[1] = (unsigned long long)k1;
    }
}
template <int NR>
__global__ __launch_bounds__(256) void select_hist_kernel(const double *__restrict__ v, int64_t n, int64_t stride, SelState *s,
                                                          int shift, int width) {
    __shared__ unsigned int h[NR][SEL_BINS];
    for (int r = 0; r < NR; ++r)
        for (int b = (int)threadIdx.x; b < SEL_BINS; b += 256) h[r][b] = 0u;
    __syncthreads();
    const int hi = shift + width;
    const unsigned long long mask = (1ULL << width) - 1ULL;
    unsigned long long pre[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) pre[r] = s->prefix[r];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long key = sel_key(v[i * stride]);
        const unsigned long long top = hi >= 64 ? 0ULL : key >> hi;
        const int bin = (int)((key >> shift) & mask);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (top == pre[r]) atomicAdd(&h[r][bin], 1u);
    }
    __syncthreads();
    for (int r = 0; r < NR; ++r)
        for (int b = (int)threadIdx.x; b < SEL_BINS; b += 256) {
            const unsigned int c = h[r][b];
            if (c) atomicAdd(&s->hist[r][b], c);
        }
}
// one workgroup of 1024 threads, two bins each: the bucket that holds the rank
__global__ __launch_bounds__(1024) void select_pick_kernel(SelState *s, int n_ranks, int width, int last, double *__restrict__ out) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int found_bin, found_before;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int r = 0; r < n_ranks; ++r) {
        const unsigned int c0 = s->hist[r][2 * t], c1 = s->hist[r][2 * t + 1];
        unsigned int inc = c0 + c1;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned int o = __shfl_up(inc, m, 64);
            if (lane >= m) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const unsigned long long k = s->k[r];
        const unsigned long long before0 = (unsigned long long)base + inc - (c0 + c1), before1 = before0 + c0;
        if (k >= before0 && k < before0 + c0) { found_bin = 2 * t; found_before = (unsigned int)before0; }
        if (k >= before1 && k < before1 + c1) { found_bin = 2 * t + 1; found_before = (unsigned int)before1; }
        s->hist[r][2 * t] = 0u; s->hist[r][2 * t + 1] = 0u;
        __syncthreads();
        if (t == 0) {
            const unsigned long long p = (s->prefix[r] << width) | (unsigned long long)found_bin;
            s->prefix[r] = p;
            s->k[r] = k - found_before;
            if (last) out[r] = sel_value(p);
        }
        __syncthreads();
    }
}
}  // namespace

size_t select_workspace_bytes() { return (sizeof(SelState) + 255) / 256 * 256; }

int select_ranks_f64(const double *values, int64_t n, int64_t stride, int n_ranks, const int64_t *ranks_host, double *out_dev,
                     void *workspace, hipStream_t st) {
    if (!values || n <= 0 || stride < 1 || n_ranks < 1 || n_ranks > SELECT_MAX_RANKS || !ranks_host || !out_dev || !workspace)
        return F4L_EINVAL;
    for (int r = 0; r < n_ranks; ++r)
        if (ranks_host[r] < 0 || ranks_host[r] >= n) return F4L_EINVAL;
    SelState *s = (SelState *)workspace;
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(1024), 0, st, s, n_ranks, (long long)ranks_host[0],
                       (long long)ranks_host[n_ranks > 1 ? 1 : 0]);
    int64_t blocks = (n + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    static const int widths[6] = {11, 11, 11, 11, 10, 10};
    int shift = 64;
    for (int p = 0; p < 6; ++p) {
        shift -= widths[p];
        if (n_ranks == 1) hipLaunchKernelGGL(select_hist_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, values, n, stride, s, shift, widths[p]);
        else hipLaunchKernelGGL(select_hist_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, values, n, stride, s, shift, widths[p]);
        hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(1024), 0, st, s, n_ranks, widths[p], p == 5 ? 1 : 0, out_dev);
    }
    F4L_LAUNCH_CHECK();
    return F4L_OK;
}
}  // namespace f4l
