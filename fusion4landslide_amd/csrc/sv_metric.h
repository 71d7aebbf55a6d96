// sv_metric.h -- the VCCS point metric of the supervoxel partition on the device.
#pragma once
#include "f4l_device.h"

namespace f4l {
// supervoxel.cpp:27-40 `VCCSMetric::operator()`:  1 - |n_a . n_b| + 0.4 |p_a - p_b| / resolution.
// The same operations in the same order as the host's Segmenter::metric (no contraction; the double sqrt and division of
// the device library are correctly rounded): bit-identical values.
__device__ __forceinline__ double sv_metric(const float *__restrict__ xyz, const double *__restrict__ nrm, int64_t a,
                                            int64_t b, double resolution) {
#pragma clang fp contract(off)
    const double dot = nrm[3 * a] * nrm[3 * b] + nrm[3 * a + 1] * nrm[3 * b + 1] + nrm[3 * a + 2] * nrm[3 * b + 2];
    const double t1 = (double)xyz[3 * a] - xyz[3 * b], t2 = (double)xyz[3 * a + 1] - xyz[3 * b + 1],
                 t3 = (double)xyz[3 * a + 2] - xyz[3 * b + 2];
    return 1.0 - fabs(dot) + sqrt(t1 * t1 + t2 * t2 + t3 * t3) / resolution * 0.4;
}
// the same value from coordinates and normals already in registers (a: float xyz + double normal, b likewise)
__device__ __forceinline__ double sv_metric_vals(const float (&pa)[3], const double (&na)[3], const float (&pb)[3], const double (&nb)[3],
                                                 double resolution) {
#pragma clang fp contract(off)
    const double dot = na[0] * nb[0] + na[1] * nb[1] + na[2] * nb[2];
    const double t1 = (double)pa[0] - pb[0], t2 = (double)pa[1] - pb[1], t3 = (double)pa[2] - pb[2];
    return 1.0 - fabs(dot) + sqrt(t1 * t1 + t2 * t2 + t3 * t3) / resolution * 0.4;
}
}  // namespace f4l
