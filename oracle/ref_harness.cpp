// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product path.
//
// Thin C-ABI driver around the *reference's own* header-only "codelibrary"
// (read in place from /root/reference/cpp_core/supervoxel_segmentation, never copied).
// The reference translation unit supervoxel.cpp is unbuildable in this image (it includes
// <pcl/io/ply_io.h>, PCL is absent), so this driver repeats its 40-line call sequence
// (reference supervoxel.cpp:92-133) with the PLY loader replaced by an in-memory float array,
// and restates the 3-line VCCSMetric functor (reference supervoxel.cpp:27-40).  Every numerical
// routine -- KDTree::FindKNearestNeighbors, PCAEstimateNormal, GridSample, Median, DisjointSet,
// SupervoxelSegmentation -- is the reference's own template code.
//
// Built by oracle/Makefile into oracle/_ref/libf4l_ref.so (git-ignored, travels with gpurun).
#include <cstdint>
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#include "codelibrary/geometry/io/xyz_io.h"
#include "codelibrary/geometry/point_cloud/pca_estimate_normals.h"
#include "codelibrary/geometry/point_cloud/supervoxel_segmentation.h"
#include "codelibrary/geometry/util/distance_3d.h"
#include "codelibrary/statistics/kernel/median.h"
#include "codelibrary/util/tree/kd_tree.h"

namespace {

struct PointWithNormal : cl::RPoint3D {
    PointWithNormal() {}
    cl::RVector3D normal;
};

// reference supervoxel.cpp:27-40
class VCCSMetric {
public:
    explicit VCCSMetric(double resolution) : resolution_(resolution) {}
    double operator()(const PointWithNormal& p1, const PointWithNormal& p2) const {
        return 1.0 - std::fabs(p1.normal * p2.normal) +
               cl::geometry::Distance(p1, p2) / resolution_ * 0.4;
    }
private:
    double resolution_;
};

}  // namespace

extern "C" {

// Runs reference supervoxel.cpp:92-133 on an in-memory cloud.
// Any output pointer may be NULL.  Returns 0, or -1 on bad arguments.
int f4l_ref_supervoxel(const float* xyz, int64_t n, int k, double resolution,
                       int32_t* knn_idx, double* knn_d2, double* normals_out,
                       int32_t* labels_out, int32_t* n_supervoxels_out,
                       int32_t* n_grid_cells_out) {
    if (!xyz || n <= 0 || k <= 0 || k >= n || !(resolution > 0.0)) return -1;
    cl::Array<cl::RPoint3D> points;
    for (int64_t i = 0; i < n; ++i) {
        double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];  // supervoxel.cpp:66-81
        points.emplace_back(x, y, z);
    }
    int n_points = points.size();

    cl::KDTree<cl::RPoint3D> kdtree;
    kdtree.SwapPoints(&points);

    cl::Array<cl::RVector3D> normals(n_points);
    cl::Array<cl::Array<int> > neighbors(n_points);
    cl::Array<cl::RPoint3D> neighbor_points(k);
    for (int i = 0; i < n_points; ++i) {
        kdtree.FindKNearestNeighbors(kdtree.points()[i], k, &neighbors[i]);
        for (int j = 0; j < k; ++j) neighbor_points[j] = kdtree.points()[neighbors[i][j]];
        cl::geometry::point_cloud::PCAEstimateNormal(neighbor_points.begin(), neighbor_points.end(),
                                                     &normals[i]);
    }
    kdtree.SwapPoints(&points);

    if (knn_idx || knn_d2) {
        for (int i = 0; i < n_points; ++i)
            for (int j = 0; j < k; ++j) {
                int q = neighbors[i][j];
                if (knn_idx) knn_idx[(int64_t)i * k + j] = q;
                if (knn_d2) knn_d2[(int64_t)i * k + j] = cl::geometry::SquaredDistance(points[i], points[q]);
            }
    }
    if (normals_out)
        for (int i = 0; i < n_points; ++i) {
            normals_out[3 * i] = normals[i].x;
            normals_out[3 * i + 1] = normals[i].y;
            normals_out[3 * i + 2] = normals[i].z;
        }

    cl::Array<PointWithNormal> oriented_points(n_points);
    for (int i = 0; i < n_points; ++i) {
        oriented_points[i].x = points[i].x;
        oriented_points[i].y = points[i].y;
        oriented_points[i].z = points[i].z;
        oriented_points[i].normal = normals[i];
    }

    if (n_grid_cells_out) {
        cl::Array<int> sampling;
        cl::geometry::point_cloud::GridSample(oriented_points.begin(), oriented_points.end(),
                                              resolution, &sampling);
        *n_grid_cells_out = sampling.size();
    }

    VCCSMetric metric(resolution);
    cl::Array<int> labels, supervoxels;
    cl::geometry::point_cloud::SupervoxelSegmentation(oriented_points, neighbors, resolution, metric,
                                                      &supervoxels, &labels);
    if (n_supervoxels_out) *n_supervoxels_out = supervoxels.size();
    if (labels_out)
        for (int i = 0; i < n_points; ++i) labels_out[i] = labels[i];
    return 0;
}

// Reference segmentation only (supervoxel_segmentation.h:254-265) on caller-provided
// neighbours + normals: lets tests feed GPU-produced kNN/normals to the reference segmenter.
int f4l_ref_segment(const float* xyz, const double* normals_in, const int32_t* knn_idx, int64_t n,
                    int k, double resolution, int32_t* labels_out, int32_t* n_supervoxels_out) {
    if (!xyz || !normals_in || !knn_idx || n <= 0 || k <= 0) return -1;
    int n_points = (int)n;
    cl::Array<PointWithNormal> oriented_points(n_points);
    cl::Array<cl::Array<int> > neighbors(n_points);
    for (int i = 0; i < n_points; ++i) {
        oriented_points[i].x = xyz[3 * i];
        oriented_points[i].y = xyz[3 * i + 1];
        oriented_points[i].z = xyz[3 * i + 2];
        oriented_points[i].normal = cl::RVector3D(normals_in[3 * i], normals_in[3 * i + 1], normals_in[3 * i + 2]);
        neighbors[i].resize(k);
        for (int j = 0; j < k; ++j) neighbors[i][j] = knn_idx[(int64_t)i * k + j];
    }
    VCCSMetric metric(resolution);
    cl::Array<int> labels, supervoxels;
    cl::geometry::point_cloud::SupervoxelSegmentation(oriented_points, neighbors, resolution, metric,
                                                      &supervoxels, &labels);
    if (n_supervoxels_out) *n_supervoxels_out = supervoxels.size();
    if (labels_out)
        for (int i = 0; i < n_points; ++i) labels_out[i] = labels[i];
    return 0;
}

// The starting lambda of the fusion (reference supervoxel_segmentation.h:105-113) -- a local variable there, so the five
// lines are repeated here on caller-provided neighbours + normals: the per-point minimum of the metric over the neighbour
// list (self skipped) and the reference's own cl::Median (statistics/kernel/median.h:22-31, nth_element at size / 2).
int f4l_ref_lambda0(const float* xyz, const double* normals_in, const int32_t* knn_idx, int64_t n, int k,
                    double resolution, double* lambda0_out) {
    if (!xyz || !normals_in || !knn_idx || n <= 0 || k <= 0 || !lambda0_out) return -1;
    int n_points = (int)n;
    cl::Array<PointWithNormal> points(n_points);
    for (int i = 0; i < n_points; ++i) {
        points[i].x = xyz[3 * i];
        points[i].y = xyz[3 * i + 1];
        points[i].z = xyz[3 * i + 2];
        points[i].normal = cl::RVector3D(normals_in[3 * i], normals_in[3 * i + 1], normals_in[3 * i + 2]);
    }
    VCCSMetric metric(resolution);
    cl::Array<double> dis(n_points, DBL_MAX);
    for (int i = 0; i < n_points; ++i)
        for (int t = 0; t < k; ++t) {
            int j = knn_idx[(int64_t)i * k + t];
            if (i != j) dis[i] = std::min(dis[i], metric(points[i], points[j]));
        }
    *lambda0_out = std::max(DBL_EPSILON, cl::Median(dis.begin(), dis.end()));
    return 0;
}

// The partition text file written by the reference's own header-only writer (codelibrary/geometry/io/xyz_io.h:192-221),
// driven like reference supervoxel.cpp:45-64 `WritePoints` (one colour per supervoxel from a default-seeded std::mt19937)
// on points widened to double like supervoxel.cpp:66-81.  Pins f4l_write_partition_txt byte for byte.
int f4l_ref_write_points(const char* filename, int32_t n_supervoxels, const float* xyz, const int32_t* labels_in,
                         int64_t n) {
    if (!filename || n < 0 || n_supervoxels < 0 || (n > 0 && (!xyz || !labels_in))) return -1;
    cl::Array<cl::RPoint3D> points;
    cl::Array<int> labels((int)n);
    for (int64_t i = 0; i < n; ++i) {
        points.emplace_back((double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]);
        labels[(int)i] = labels_in[i];
    }
    cl::Array<cl::RGB32Color> colors(points.size());
    std::mt19937 random;
    cl::Array<cl::RGB32Color> supervoxel_colors(n_supervoxels);
    for (int i = 0; i < n_supervoxels; ++i) supervoxel_colors[i] = cl::RGB32Color(random());
    for (int i = 0; i < points.size(); ++i) colors[i] = supervoxel_colors[labels[i]];
    return cl::geometry::io::WriteXYZPoints(filename, points, colors, labels) ? 0 : -2;
}

// Reference PCAEstimateNormal (pca_estimate_normals.h:118-121) on one neighbourhood of m points.
int f4l_ref_pca_normal(const double* pts, int m, double* normal_out) {
    if (!pts || m <= 0 || !normal_out) return -1;
    cl::Array<cl::RPoint3D> nb(m);
    for (int i = 0; i < m; ++i) nb[i] = cl::RPoint3D(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    cl::RVector3D nrm;
    cl::geometry::point_cloud::PCAEstimateNormal(nb.begin(), nb.end(), &nrm);
    normal_out[0] = nrm.x; normal_out[1] = nrm.y; normal_out[2] = nrm.z;
    return 0;
}

}  // extern "C"
