// select.h -- order statistics of doubles on the device without sorting them (csrc/select.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace f4l {
constexpr int SELECT_MAX_RANKS = 2;
// Bytes of device workspace select_ranks_f64 needs (independent of n).
size_t select_workspace_bytes();
// out_dev[r] = the element of rank ranks_host[r] (0-based, ascending) among the n doubles values[i * stride]; 1 <= n_ranks <=
// SELECT_MAX_RANKS, no NaNs.  Enqueues on `st`, never synchronises.  Returns an F4L_* status.
int select_ranks_f64(const double *values, int64_t n, int64_t stride, int n_ranks, const int64_t *ranks_host, double *out_dev,
                     void *workspace, hipStream_t st);
}  // namespace f4l
