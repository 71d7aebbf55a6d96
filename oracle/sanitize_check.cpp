// TEST INFRASTRUCTURE ONLY -- sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer, CPU only) of
//   * the C oracle (oracle/f4l_oracle.c): kNN, PCA normals, supervoxel segmentation, Kabsch, ICP, nearest neighbours;
//   * the HOST code of the product library that is not a GPU kernel: the label-identical sequential segmentation
//     (f4l_supervoxel_segment_host, fusion4landslide_amd/csrc/supervoxel.hip) and the partition text writer
//     (f4l_write_partition_txt), compiled for the host only.
// The two segmentations must give identical labels on a seeded cloud; any sanitizer report aborts with a non-zero exit.
// Built and run by `make -C oracle asan` (the reference has no sanitizer builds at all).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
int orc_knn(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out);
int orc_normals_from_knn(const float *xyz, int64_t n, const int32_t *knn_idx, int k, double *normals_out);
int orc_supervoxel_segment(const float *xyz, const double *normals, const int32_t *knn_idx, int64_t n, int k, double resolution,
                           int32_t *labels_out, int32_t *n_supervoxels_out, double *lambda0_out);
int orc_kabsch_batched(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P, double w_thresh, double eps,
                       double *R_out, double *t_out);
int orc_piecewise_icp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P, const double *init_T,
                      double max_corr_dist, int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters, double *T_out,
                      double *fitness_out, double *rmse_out, int32_t *iters_out);
int orc_nn_within(const double *query, int64_t nq, const double *tgt, int64_t nt, double thr, int32_t *idx_out, double *d2_out);
int f4l_supervoxel_segment_host(const float *xyz_host, const double *normals_host, const int32_t *knn_host, int64_t n, int k,
                                double resolution, int32_t *labels_host, int32_t *n_supervoxels_host);
int f4l_write_partition_txt(const char *path, const float *xyz_host, const int32_t *labels_host, int64_t n, int32_t n_supervoxels);
}

static unsigned long long rng_state = 88172645463325252ULL;
static double uni() {  // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
    const int n = 6000, k = 20;
    const double res = 0.25;
    std::vector<float> xyz(3 * n), tgt(3 * n);
    for (int i = 0; i < n; ++i) {
        const double x = 2.0 * uni(), y = 2.0 * uni();
        xyz[3 * i] = (float)x; xyz[3 * i + 1] = (float)y; xyz[3 * i + 2] = (float)(0.1 * std::sin(5 * x) * std::cos(4 * y) + 0.002 * (uni() - 0.5));
        tgt[3 * i] = xyz[3 * i] + 0.01f; tgt[3 * i + 1] = xyz[3 * i + 1] - 0.02f; tgt[3 * i + 2] = xyz[3 * i + 2] + 0.005f;
    }
    std::vector<int32_t> idx((size_t)n * k), lab_o(n), lab_h(n);
    std::vector<double> d2((size_t)n * k), nrm(3 * n);
    int rc = orc_knn(xyz.data(), n, k, idx.data(), d2.data());
    if (rc) { std::fprintf(stderr, "orc_knn %d\n", rc); return 1; }
    rc = orc_normals_from_knn(xyz.data(), n, idx.data(), k, nrm.data());
    if (rc) { std::fprintf(stderr, "orc_normals %d\n", rc); return 1; }
    int32_t K_o = 0, K_h = 0;
    double lambda0 = 0;
    rc = orc_supervoxel_segment(xyz.data(), nrm.data(), idx.data(), n, k, res, lab_o.data(), &K_o, &lambda0);
    if (rc) { std::fprintf(stderr, "orc_supervoxel_segment %d\n", rc); return 1; }
    rc = f4l_supervoxel_segment_host(xyz.data(), nrm.data(), idx.data(), n, k, res, lab_h.data(), &K_h);
    if (rc) { std::fprintf(stderr, "f4l_supervoxel_segment_host %d\n", rc); return 1; }
    if (K_o != K_h) { std::fprintf(stderr, "supervoxel counts differ: %d vs %d\n", K_o, K_h); return 2; }
    for (int i = 0; i < n; ++i)
        if (lab_o[i] != lab_h[i]) { std::fprintf(stderr, "labels differ at %d\n", i); return 2; }
    const char *out = argc > 1 ? argv[1] : "/tmp/f4l_sanitize_partition.txt";
    rc = f4l_write_partition_txt(out, xyz.data(), lab_h.data(), n, K_h);
    if (rc) { std::fprintf(stderr, "f4l_write_partition_txt %d\n", rc); return 1; }
    // Kabsch + ICP + nearest neighbours of the oracle on ragged patches (one empty)
    const int64_t off[5] = {0, 1500, 1500, 4000, n};
    std::vector<double> R(9 * 4), t(3 * 4), T(16 * 4), fit(4), rmse(4);
    std::vector<int32_t> iters(4);
    rc = orc_kabsch_batched(xyz.data(), tgt.data(), nullptr, off, 4, 0.0, 1e-6, R.data(), t.data());
    if (rc) { std::fprintf(stderr, "orc_kabsch_batched %d\n", rc); return 1; }
    rc = orc_piecewise_icp(xyz.data(), off, tgt.data(), off, 4, nullptr, 0.1, 10, 1e-6, 1e-6, 0, 0, T.data(), fit.data(), rmse.data(), iters.data());
    if (rc) { std::fprintf(stderr, "orc_piecewise_icp %d\n", rc); return 1; }
    std::vector<double> q(300), tg(3 * 500);
    for (double &v : q) v = uni();
    for (double &v : tg) v = uni();
    std::vector<int32_t> nn(100);
    std::vector<double> nd(100);
    rc = orc_nn_within(q.data(), 100, tg.data(), 500, 0.2, nn.data(), nd.data());
    if (rc) { std::fprintf(stderr, "orc_nn_within %d\n", rc); return 1; }
    std::printf("sanitize_check ok: n=%d K=%d lambda0=%.6g fitness[0]=%.3f\n", n, K_h, lambda0, fit[0]);
    return 0;
}
