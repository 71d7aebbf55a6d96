// supervoxel_host.h -- host-side segmentation shared by supervoxel_host.cpp (its body) and supervoxel.hip (its caller).
#pragma once
#include <stdint.h>

namespace f4l {
// supervoxel_segmentation.h:65-265.  Returns the number of supervoxels, or a negative F4L_E* code.
// The two embarrassingly parallel sweeps of the segmentation can be computed elsewhere (f4l_supervoxel: on the GPU,
// bit-identically) and handed in: dis0[i] = smallest metric from point i to a neighbour (:105-113), and, once the
// fusion has produced the labels, flag[i] = "some neighbour of i carries another label" together with
// dis[i] = metric(i, its representative) (:186-200).  Null members: computed here on the host.
struct SegmentAssist {
    const double *dis0 = nullptr;
    void (*boundary)(const int32_t *labels, uint8_t *flag, double *dis, void *ctx) = nullptr;
    void *ctx = nullptr;
};
int segment_host(const float *xyz, const double *nrm, const int32_t *knn, int64_t n64, int k, double resolution, int32_t *labels,
                 const SegmentAssist &assist = SegmentAssist());
}  // namespace f4l
