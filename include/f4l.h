/*
 * f4l.h -- C ABI of the MI355X-native piecewise-ICP displacement-field engine (libf4l_hip.so).
 *
 * This is the drop-in boundary for the hot path of gseg-ethz/fusion4landslide.  Each entry point
 * names the reference interface it replaces (file:line relative to the reference checkout).  The
 * reference has no FFI of its own on this path except the SWIG module `supervoxel`
 * (cpp_core/supervoxel_segmentation/supervoxel.i:15,22); the other boundaries are Python call sites
 * (utils/o3d_tools.py:12, scripts/weighted_svd.py:58,132) which the host-side mirror in
 * fusion4landslide_amd/ re-implements on top of these functions via ctypes.
 *
 * Conventions
 *  - All array arguments are DEVICE pointers (HBM resident, e.g. torch tensor .data_ptr()) unless the
 *    parameter name ends in `_host`.  The caller owns every buffer.
 *  - Point clouds are packed float32 [n][3] (x,y,z), exactly the (N,3) tensors the reference keeps
 *    (src/coarse_to_fine_matching_base.py:906-912).  Ragged patches are CSR: `off` is int64 [P+1].
 *  - Transforms are row-major double [4][4]; rotations row-major double [3][3]  (Open3D returns
 *    float64 4x4, utils/o3d_tools.py:66).
 *  - `stream` is a hipStream_t (NULL = default stream).  Work is enqueued asynchronously on it; no
 *    call synchronises the device unless documented.  Re-entrant; the only state kept between calls is
 *    a per-device set of helper streams (f4l_piecewise_icp / f4l_patch_loop run the size classes of an uneven
 *    batch side by side on them; they fork from and join `stream` through events, so the caller sees one stream).
 *  - Return value: F4L_OK (0) or a negative F4L_E* code; nothing throws.
 *  - Functions taking `workspace` need a caller-provided scratch buffer of at least the byte count the
 *    matching *_workspace_bytes() query returns (so that nothing allocates inside a launch and the
 *    sequence can be captured into a hipGraph).
 */
#ifndef F4L_H_
#define F4L_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F4L_OK 0
#define F4L_EINVAL (-1)       /* bad argument (NULL pointer, negative size, k out of range ...) */
#define F4L_EWORKSPACE (-2)   /* workspace too small */
#define F4L_EHIP (-3)         /* HIP runtime error, see f4l_last_hip_error() */
#define F4L_EUNSUPPORTED (-4) /* valid request this build cannot serve (e.g. k > 64) */
#define F4L_ENOMEM (-5)       /* host allocation failed */

#define F4L_ICP_POINT2POINT 0 /* o3d TransformationEstimationPointToPoint(False), utils/o3d_tools.py:34 */
#define F4L_ICP_POINT2PLANE 1 /* o3d TransformationEstimationPointToPlane(),     utils/o3d_tools.py:39 */
#define F4L_ICP_GENERALIZED 2 /* o3d TransformationEstimationForGeneralizedICP,    utils/o3d_tools.py:41 (f4l_piecewise_gicp only) */
/* Two semantics of the point-to-plane STEP, which differ only where the 6 x 6 system does not pin its six unknowns:
 *   default (ROBUST, this library's own rule): a step with fewer than six correspondences, or whose system is singular to
 *     1e-13 of its largest diagonal entry, is not taken -- the transform stays and the loop ends on its criteria.  The default
 *     of the batched entry points because ONE rank-deficient patch of a 100 k-patch launch otherwise leaves with a transform
 *     made of rounding noise (a four-pair patch was thrown 55 m, tools/gpu/fuzz_icp.py 1 2250095 f64 n32).
 *   F4L_ICP_P2PL_OPEN3D, OR-ed into `mode`: Open3D's own semantics, what the reference's call
 *     utils/o3d_tools.py:38-39,46-50 does -- TransformationEstimationPointToPlane::ComputeTransformation ->
 *     SolveJacobianSystemAndObtainExtrinsicMatrix -> SolveLinearSystemPSD with its checks off: x = JTJ.ldlt().solve(-JTr) in
 *     the CALLER's frame (Eigen's diagonal-pivoted L D L^T, pseudo-inverse of D), applied whenever there is at least ONE
 *     correspondence.  What the host-side mirror of `icp_registration` passes (a drop-in does what the reference does).
 *     Only a step whose solution is not finite is left out (Open3D would carry the NaNs on).
 *   On patches that pin their six unknowns the two are the same minimiser (tested to 1e-7 m). */
#define F4L_ICP_P2PL_OPEN3D 0x400
/* f4l_patch_loop only, OR-ed into `mode`: the Kabsch transform is rounded to float32 before ICP starts from it, as the
 * reference hands Open3D the float32 4 x 4 of refine_local_rigid_correspondences (scripts/weighted_svd.py:148-151,
 * src/coarse_to_fine_matching_base.py:3360 `initial_transform=est_transform_svd.cpu()`). */
#define F4L_ICP_INIT_ROUND_F32 0x100
#define F4L_ICP_NORMALS_F64 0x200 /* tgt_normals points at doubles */

#define F4L_SEARCH_F32 0 /* nearest-neighbour search in float32 on patch-relative coordinates (fast path)   */
#define F4L_SEARCH_F64 1 /* ... in float64, the arithmetic of the reference's Open3D path (parity mode)     */

#define F4L_MAX_K 64 /* neighbour-list capacity of the wave-resident top-k (one slot per lane) */

int f4l_version(void);
const char *f4l_strerror(int code);
/* hipError_t of the most recent failing HIP call on this thread (0 when none). */
int f4l_last_hip_error(void);
/* CU count, LDS bytes per workgroup, total HBM bytes, gcnArchName of the current device. */
int f4l_device_info(int *cu_count, int *lds_bytes, int64_t *hbm_bytes, char *arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * B3  weighted Kabsch / Procrustes, batched over ragged correspondence lists.
 * Replaces scripts/weighted_svd.py:58-129 `weighted_procrustes` (dup src/rgb_guided.py:25-96) as it is
 * called once per patch at src/coarse_to_fine_matching_base.py:3341 (via :132-142):
 *   w <- where(w < w_thresh, 0, w);  w <- w / (sum w + eps)      (eps stays in the denominator)
 *   cs, ct weighted centroids;  H = sum w (s - cs)(t - ct)^T;  U S V^T = svd(H)
 *   R = V diag(1,1,sign det(V U^T)) U^T;  t = ct - R cs
 * src/ref: float32 [n_total][3]; w: float32 [n_total] or NULL (all ones); off: int64 [P+1].
 * R_out: double [P][9]; t_out: double [P][3].  A patch with zero rows yields R = I, t = 0.
 * The _f64 variant takes double clouds/weights (the reference function is dtype generic).
 * ---------------------------------------------------------------------------------------------- */
int f4l_kabsch_batched(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                       int64_t n_total, double w_thresh, double eps, double *R_out, double *t_out,
                       void *stream);
int f4l_kabsch_batched_f64(const double *src, const double *ref, const double *w, const int64_t *off,
                           int64_t P, int64_t n_total, double w_thresh, double eps, double *R_out,
                           double *t_out, void *stream);

/* The same solve with `return_transform=True` (scripts/weighted_svd.py:115-120): T_out double [P][16], row-major
 * 4x4 [R t; 0 0 0 1] per patch -- the `init` of the ICP call that follows at
 * src/coarse_to_fine_matching_base.py:3358, without any glue kernels in between. */
int f4l_kabsch_transforms(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                          int64_t n_total, double w_thresh, double eps, double *T_out, void *stream);

/* Kabsch #2: src/functions.py:12-85 `kabsch_transformation_estimation` (the F2S3 / outlier-classifier variant;
 * callers src/f2s3.py:340-366, src/models/outlier_classifier.py:65-106), batched over ragged correspondence lists:
 *   w <- w / (sum w + eps) when normalize_w;  w[w < w_thresh] <- 0 when w_thresh > 0;
 *   means divided by (sum w + eps) again;  H = sum (x1 - m1) w (x2 - m2)^T;  U S V^T = svd(H);
 *   R = V diag(1, 1, det(V U^T)) U^T  (the determinant itself, as the reference);  t = m2 - R m1.
 * `best_k` of the reference (which applies batch element 0's selection to every element) is not offered. */
int f4l_kabsch2_batched(const float *src, const float *ref, const float *w, const int64_t *off, int64_t P,
                        int64_t n_total, int normalize_w, double w_thresh, double eps, double *R_out, double *t_out,
                        void *stream);
int f4l_kabsch2_batched_f64(const double *src, const double *ref, const double *w, const int64_t *off, int64_t P,
                            int64_t n_total, int normalize_w, double w_thresh, double eps, double *R_out,
                            double *t_out, void *stream);

/* Residual norms || R_p s_i + t_p - r_i || per row (scripts/weighted_svd.py:143-146), float64 [n_total].
 * The caller prunes rows (res < 1 m at :147, or 2.5 x median at src/rgb_guided.py:113-118). */
int f4l_kabsch_residuals(const float *src, const float *ref, const int64_t *off, int64_t P, int64_t n_total,
                         const double *R, const double *t, double *res_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * B2  per-patch ICP, batched: the loop body of src/coarse_to_fine_matching_base.py:3353-3367, i.e.
 * utils/o3d_tools.py:12-71 `icp_registration` -> Open3D 0.19 registration_icp, for P independent
 * (source patch, target patch) pairs in one launch, with zero host round trips.
 *   src/src_off, tgt/tgt_off : CSR patches (float32 [.][3], int64 [P+1])
 *   init_T      : double [P][16] or NULL (identity)              (o3d `init`, utils/o3d_tools.py:47)
 *   tgt_normals : float32 [n_tgt][3] -- or, with F4L_ICP_NORMALS_F64 or-ed into `mode`, double [n_tgt][3] passed through the same
 *                 pointer --; required for POINT2PLANE (see f4l_patch_normals / f4l_patch_normals_f64), else NULL
 *   max_corr_dist, max_iter, rel_fitness, rel_rmse : o3d ICPConvergenceCriteria (utils/o3d_tools.py:47-50)
 *   fixed_iters != 0 disables the early exit: exactly max_iter updates (benchmark mode)
 *   search_precision : F4L_SEARCH_F32 | F4L_SEARCH_F64 (F4L_ICP_POINT2PLANE always runs F64).  The transform, all sums and the solves are double in
 *                 both; F64 also evaluates point positions and squared distances in double, like Open3D.
 *   max_src_patch_host / max_tgt_patch_host : largest patch sizes (host-known; size LDS and pick the path)
 *   n_src_host  : src_off[P] if the host knows it, else 0.  With it the launcher sees whether patch sizes are uneven
 *                 (mean well below the largest) and then bins the patches by size on the device, so that patches
 *                 of one or two wavefronts get workgroups of just those; results do not depend on it.
 * Outputs (any may be NULL except T_out):
 *   T_out double [P][16]; fitness_out, rmse_out double [P]; iters_out int32 [P];
 *     iters_out[p] < 0 flags a patch that did not iterate normally: -1 skipped (f4l_patch_loop: fewer than min_corr pairs),
 *     -2 a step with Open3D's semantics (F4L_ICP_P2PL_OPEN3D, generalized ICP) was not finite -- Open3D would return a NaN
 *     transform there (generalized ICP with epsilon = 0 on a pair of exactly parallel normals); the patch keeps its last
 *     finite transform and stops.  max_*_patch_host may be understated: larger patches only take a slower path.
 *   corr_out int32 [n_src]: index INSIDE the target patch of each source point's final correspondence,
 *   or -1 (utils/o3d_tools.py:64 `correspondence_set`).
 * ---------------------------------------------------------------------------------------------- */
int f4l_piecewise_icp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off,
                      int64_t P, const double *init_T, const float *tgt_normals, double max_corr_dist,
                      int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                      int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host,
                      int64_t n_src_host, double *T_out, double *fitness_out, double *rmse_out, int32_t *iters_out,
                      int32_t *corr_out, void *stream);

/* ... with icp_type 'generalized_icp' (utils/o3d_tools.py:40-41,51-56: TransformationEstimationForGeneralizedICP through
 * registration_generalized_icp), for P patch pairs in one launch.  The reference has no caller for this type; it is here so
 * that `icp_registration` is whole.  Both clouds carry normals when Open3D reaches the call (:29-30), so their covariances are
 *   C_i = Rx diag(epsilon, 1, 1) Rx^T,  Rx = GetRotationFromE1ToX(n_i)  (= I - (1 - epsilon) n_i n_i^T; e1's for n_i.x < -0.99),
 * the source's turning with the cloud; per correspondence M = C_q + R C_s R^T, and the step minimises sum d^T M^-1 d through
 * the 6 x 6 normal equations, solved and applied with Open3D's semantics (as F4L_ICP_P2PL_OPEN3D).
 *   src_normals / tgt_normals : double [n_src][3] / [n_tgt][3], unit length (f4l_patch_normals_f64 with knn = 30)
 *   epsilon     : the estimator's first parameter.  The reference passes `False` there, i.e. 0.0 (Open3D's default: 1e-3);
 *                 with 0 a pair of exactly parallel normals makes M singular -- Open3D divides by zero there; this kernel
 *                 leaves out a step that is not finite
 *   everything else as f4l_piecewise_icp (the search runs in float64). */
int f4l_piecewise_gicp(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                       const double *init_T, const double *src_normals, const double *tgt_normals, double epsilon,
                       double max_corr_dist, int max_iter, double rel_fitness, double rel_rmse, int fixed_iters,
                       int64_t max_src_patch_host, int64_t max_tgt_patch_host, int64_t n_src_host, double *T_out,
                       double *fitness_out, double *rmse_out, int32_t *iters_out, int32_t *corr_out, void *stream);

/* The loop body's steps before the rigid fit, batched over P patch matches.
 *
 * f4l_mutual_correspondences (src/coarse_to_fine_matching_base.py:3259-3274): a patch match i is (source patch = point ids
 *   src_ids[src_off[i]:src_off[i+1]], target patch = tgt_ids[tgt_off[i]:tgt_off[i+1]], ids ascending inside a target
 *   patch, as f4l_labels_to_csr emits them); corr_tgt int64 [n_corr] holds the target point matched to every source point
 *   (-1: none; the second column of `corres_3d_voxel_from_3d_idx`).  mask_out uint8 [src_off[P]] = 1 where the source
 *   point's correspondent lies in the matched target patch (the `torch.isin` of :3260); count_out int64 [P] (nullable) =
 *   mutual pairs per match.  Compacting the rows the mask keeps gives the CSR correspondence lists of f4l_patch_loop.
 * f4l_rigidity_check (:3304-3320, `remove_low_quality_patch_matches`): per match, over its n mutual pairs,
 *   dist_mean = mean over i < j of |d(s_i, s_j) - d(t_i, t_j)|,  ratio_inlier = share of pairs with that difference <=
 *   thres_dist_diff (the reference counts both triangles and takes the diagonal off again: the same ratio); both 0 for
 *   n < 2.  The caller drops a match when ratio_inlier <= thres_inlier_ratio or dist_mean >= thres_dist_diff (:3322).
 *   Distances in double (the reference: float32 torch.cdist).  f4l_rigidity_check_f32: the same test with the pair
 *   arithmetic in float32 -- coordinate differences inside a patch are exact there, a distance is within 3e-7 of itself
 *   (closer than the reference's own float32 cdist) -- several times faster; match sets beyond 1024 pairs run in double. */
int f4l_mutual_correspondences(const int64_t *src_ids, const int64_t *src_off, const int64_t *tgt_ids,
                               const int64_t *tgt_off, int64_t P, const int64_t *corr_tgt, int64_t n_corr,
                               uint8_t *mask_out, int64_t *count_out, void *stream);
int f4l_rigidity_check(const float *corr_src, const float *corr_ref, const int64_t *corr_off, int64_t P,
                       double thres_dist_diff, double *dist_mean_out, double *ratio_inlier_out, void *stream);
int f4l_rigidity_check_f32(const float *corr_src, const float *corr_ref, const int64_t *corr_off, int64_t P,
                           double thres_dist_diff, double *dist_mean_out, double *ratio_inlier_out, void *stream);

/* The rest of the loop body of src/coarse_to_fine_matching_base.py:3254-3436 for P patch matches in ONE launch:
 *   weighted Kabsch of the match's correspondences (:3341, scripts/weighted_svd.py:58-129; corr_src / corr_ref float32
 *   [n_corr][3], corr_w float32 [n_corr] or NULL, corr_off int64 [P+1])
 *   ->  ICP as f4l_piecewise_icp from that transform (:3353-3367) on the clouds src / tgt.  For parity with the reference
 *       these are the MUTUAL points of the match (`tensor2pcd(pts_coord_in_curr_spt_src_mutual)`, :3352-3353), i.e.
 *       src = corr_src and tgt = corr_ref with src_off = tgt_off = corr_off; any other per-patch clouds are accepted.
 *   ->  displacement rows [s, T s] (:3371-3374, 3408) of rows_src / rows_off: ALL points of the source patch
 *       (`pts_coord_in_curr_spt_src`, :3348); rows_src = rows_off = NULL writes the rows of the ICP cloud `src` itself.
 *       rows_out float32 [rows_off[P] or src_off[P]][6], nullable.
 *   A match with fewer than min_corr correspondences is skipped like the reference skips it (:3338 `num_min_fine_match`,
 *   `mask_spt_match_global[i] = False`): T = identity, fitness = rmse = 0, iters = -1, correspondences -1, its rows are
 *   NOT written.  (min_corr = 0: a match without correspondences starts ICP from the identity.)
 * Same results as f4l_kabsch_transforms -> f4l_piecewise_icp -> f4l_apply_transform (T_out to rounding of the block
 * reductions, rows_out bit-equal to f4l_apply_transform applied to T_out), without the two extra launches. */
int f4l_patch_loop(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                   const float *corr_src, const float *corr_ref, const float *corr_w, const int64_t *corr_off,
                   int64_t min_corr, double kabsch_w_thresh, double kabsch_eps, const float *tgt_normals,
                   double max_corr_dist, int max_iter, double rel_fitness, double rel_rmse, int mode, int fixed_iters,
                   int search_precision, int64_t max_src_patch_host, int64_t max_tgt_patch_host, int64_t n_src_host,
                   double *T_out, double *fitness_out, double *rmse_out, int32_t *iters_out, int32_t *corr_out,
                   const float *rows_src, const int64_t *rows_off, float *rows_out, void *stream);

/* Per-patch normal estimation as utils/o3d_tools.py:29-30 (`pcd.estimate_normals()` on the patch cloud:
 * kNN(knn=30) inside the patch, self included; smallest-eigenvector of the neighbourhood covariance;
 * (0,0,1) when degenerate; unoriented).  normals_out float32 [n][3]. */
int f4l_patch_normals(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                      float *normals_out, void *stream);
/* ... as doubles, normals_out double [n][3]: Open3D keeps normals in double and `registration_icp` reads them so; hand them to
 * f4l_piecewise_icp / f4l_patch_loop with F4L_ICP_NORMALS_F64 in `mode` (float32 normals move some point-to-plane results by
 * millimetres: tools/gpu/fuzz_icp.py). */
int f4l_patch_normals_f64(const float *pts, const int64_t *off, int64_t P, int knn, int64_t max_patch_host,
                          double *normals_out, void *stream);

/* a14: dense displacement rows [s, T_p s] for every point of every patch
 * (src/coarse_to_fine_matching_base.py:3371-3374,3408); out6 float32 [n][6].
 * inverse != 0 writes [T_p^-1 q, q] instead (tgt2src, :3393-3397). */
int f4l_apply_transform(const float *pts, const int64_t *off, int64_t P, int64_t n_total, const double *T,
                        int inverse, float *out6, void *stream);

/* a15: `refine_dvfs_with_threshold` (src/coarse_to_fine_matching_base.py:48-97) for all patches at once:
 * for each source point, transform by T_p, take the nearest point of the target patch; valid when
 * d^2 < thr_p^2.  thr double [P].  nn_out int32 [n_src] (index inside the target patch, or -1);
 * out6 float32 [n_src][6] = [s, nearest target] (rows of invalid points are zero filled); both nullable.
 * max_tgt_patch_host sizes LDS and picks the path; it may be understated (a larger patch streams its targets from
 * global memory: slower, same answers) -- every row of every patch is written either way. */
int f4l_nn_refine(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                  const double *T, const double *thr, int64_t max_tgt_patch_host, int32_t *nn_out, float *out6,
                  void *stream);

/* f4l_nn_refine's answers as the CSR correspondence lists f4l_patch_loop takes (the mutual pairs of every patch match,
 * src/coarse_to_fine_matching_base.py:3259-3274 when the point matches are nearest neighbours inside the matched patch): for
 * every source row i with nn[i] >= 0, in row order, corr_src_out gets src[i] and corr_ref_out the target row
 * tgt_off[patch of i] + nn[i]; corr_off_out int64 [P + 1] the lists' offsets.  kept_before int64 [src_off[P] + 1]: the number of
 * rows j < i with nn[j] >= 0 (an exclusive running count with the total behind it: one scan, the caller's); corr_*_out float32
 * [kept_before[src_off[P]]][3]. */
int f4l_match_lists(const float *src, const int64_t *src_off, const float *tgt, const int64_t *tgt_off, int64_t P,
                    const int32_t *nn, const int64_t *kept_before, float *corr_src_out, float *corr_ref_out,
                    int64_t *corr_off_out, void *stream);

/* Tile preparation around the hot loop, `_voxel_subsampling` (src/coarse_to_fine_matching_base.py:1012-1057).
 *
 * f4l_voxel_downsample: Open3D `PointCloud.voxel_down_sample(voxel)` as called at :1024-1025 -- voxel index =
 *   floor((p - (min_bound - voxel/2)) / voxel) in double, one output point per occupied voxel = the mean of its
 *   points (summed in ascending input index).  Voxels come out in ascending (z, y, x) index order (Open3D's own
 *   order is that of a hash map: unpinned).  pts_out double [n][3] (room for n, the first *m_out_host rows are
 *   written); count_out int32 [n] (points per voxel) and voxel_of_point_out int32 [n] (voxel of every input
 *   point) nullable; m_out_host is a HOST int64.  Synchronises `stream`.
 * f4l_nn_query: the k nearest points of `cloud` for every query point (the cKDTree(...).query(sub, k=1) of
 *   :1042-1046), squared Euclidean distance in double, ascending, exact-distance ties ordered by cloud index.
 *   idx_out int32 [m][k]; d2_out double [m][k] or NULL.  1 <= k <= 64, k <= n.  Synchronises `stream`. */
#define F4L_VOXEL_OPEN3D 0 /* cells anchored at min_bound - voxel/2, index by division in double (Open3D) */
#define F4L_VOXEL_PCL 1    /* cells floor(p * (1/leaf)) in float32, counted from floor(min * (1/leaf)) (pcl::VoxelGrid, the
                              filter of cpp_core/pcd_tiling/pcd_tiling.cpp:118-227); same output order, centroids in double */
size_t f4l_voxel_downsample_workspace_bytes(int64_t n);
int f4l_voxel_downsample(const float *xyz, int64_t n, double voxel, int layout, double *pts_out, int32_t *count_out,
                         int32_t *voxel_of_point_out, int64_t *m_out_host, void *workspace, size_t workspace_bytes,
                         void *stream);
size_t f4l_nn_query_workspace_bytes(int64_t n, int64_t m, int k);
int f4l_nn_query(const float *cloud, int64_t n, const float *queries, int64_t m, int k, int32_t *idx_out,
                 double *d2_out, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * B1  supervoxel partition.  Replaces the SWIG export
 *   std::vector<int> computeSupervoxel(std::string input_file, int k_neighbors, double resolution,
 *                                      std::string save_file)     cpp_core/supervoxel_segmentation/supervoxel.h:11-12
 * (body supervoxel.cpp:83-143) minus its file I/O, which the host-side shim does.
 *
 * f4l_knn: exact k nearest neighbours of every point within the cloud, squared Euclidean distance in
 *   double, self included in slot 0, ascending (codelibrary/util/tree/kd_tree.h:266-280); exact-distance
 *   ties are ordered by point index.  idx_out int32 [n][k]; d2_out double [n][k] or NULL.  1 <= k <= 64, k <= n.
 * f4l_normals: PCA normal of each point's neighbour list (codelibrary/geometry/point_cloud/
 *   pca_estimate_normals.h:43-108, unit weights), double [n][3].
 * f4l_supervoxel: kNN + normals on the device, then the reference's sequential boundary-preserving segmentation
 *   (codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-265, metric supervoxel.cpp:27-40), label for label: on the
 *   device too since round 5 (f4l_supervoxel_segment_exact below; a one-core host replay of the same sequence is the fall-back
 *   for clouds that outgrow the device's buffers).  SYNCHRONISES `stream`.  labels_out int32 [n] (device); n_supervoxels_host
 *   is a HOST int.  knn_out / normals_out (device, nullable) receive the intermediate products.
 *   F4L_EUNSUPPORTED: the neighbour graph has more connected components than the resolution grid has occupied cells (e.g. k = 4
 *   on a thin strip): no lambda ever reaches the target count and the reference's loop (:117-176) never returns on such a cloud;
 *   this call does, with nothing written to labels_out.
 * ---------------------------------------------------------------------------------------------- */
size_t f4l_knn_workspace_bytes(int64_t n, int k);
int f4l_knn(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, void *workspace,
            size_t workspace_bytes, void *stream);
int f4l_normals(const float *xyz, int64_t n, const int32_t *knn_idx, int k, double *normals_out, void *stream);
/* f4l_knn and f4l_normals in one launch (supervoxel.cpp:105-113: the loop that finds the neighbours AND estimates the
 * normal of every point): the normal is computed while the neighbour list is still in registers, which saves the 120 B per
 * point the separate pass re-reads and its scattered gathers.  Same results as the two calls.  Synchronises `stream` once
 * (grid sizing, like f4l_knn). */
int f4l_knn_normals(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, double *normals_out,
                    void *workspace, size_t workspace_bytes, void *stream);
/* ... and with the squared distance of every point to its nearest OTHER point (slot 1 of its row; double [n], k >= 2): the
 * quantity `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754) takes the median of, so that the
 * partition's neighbour search serves the resolution estimate too (same values as f4l_knn with k = 2). */
int f4l_knn_normals_nn1(const float *xyz, int64_t n, int k, int32_t *idx_out, double *d2_out, double *normals_out,
                        double *nn1_d2_out, void *workspace, size_t workspace_bytes, void *stream);
size_t f4l_supervoxel_workspace_bytes(int64_t n, int k);
int f4l_supervoxel(const float *xyz, int64_t n, int k, double resolution, int32_t *labels_out,
                   int32_t *n_supervoxels_host, int32_t *knn_out, double *normals_out, void *workspace,
                   size_t workspace_bytes, void *stream);

/* The segmentation stage ENTIRELY ON THE DEVICE (rows a5-a7 without the host): the parallel variant of
 * codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-248 with the count of grid_sample.h:31-75.  Same structure
 * and the same criteria as the reference (K = occupied cells of the resolution grid; lambda from the median of the smallest
 * neighbour metric, doubled per round; a representative v is fused into a neighbouring representative u when
 * sizes[v] * metric(u, v) < lambda; never below K; boundary exchange until no point has a neighbour whose representative is
 * strictly closer; labels 0..K-1 in ascending order of the representative's index), but the representatives of a round are
 * fused in conflict-free parallel sub-rounds instead of one after the other, so the labels are NOT those of the
 * sequential reference (f4l_supervoxel keeps the label-identical replay); they satisfy the same invariants and are
 * deterministic.  Never synchronises `stream`, never touches host memory: a fixed schedule of launches whose trip counts
 * live on the device.
 *   xyz float32 [n][3], normals double [n][3], knn int32 [n][k] (device; e.g. from f4l_knn / f4l_normals); a knn entry
 *   equal to its own row, or negative, is "no neighbour" (how a slab of a larger cloud drops neighbours it does not own)
 *   grid_bbox_host: HOST float32 [6] = {min x, y, z, max x, y, z} anchoring the resolution grid whose occupied cells are
 *   counted, or NULL for the cloud's own bounding box (grid_sample.h:48-51); a slab passes the WHOLE cloud's box so that the
 *   slabs' counts add up to the whole cloud's
 *   labels_out int32 [n];  reps_out int32 [n] or NULL: reps_out[l] = index of supervoxel l's representative point;
 *   info_out int32 [8] (device) or NULL: {supervoxels produced, K wanted, status bits, exchange sweeps run, the fusion's
 *   starting lambda (supervoxel_segmentation.h:105-113) as the low and the high word of the double, lambda rounds entered,
 *   1 + the number of the sub-round whose proposals were cut to reach K exactly (0: none)};
 *   status bit 0: the graph of representatives ran out of edges above K (disconnected cloud; the reference would not
 *   return), bit 1: lambda schedule exhausted above K, bit 2: exchange stopped by the sweep budget before its fixed point,
 *   bit 3: one sub-round offered more than half of the cloud's n k neighbour edges at once (no real cloud does; refused, with bit 1).
 *   n must be below 2^28 (edge keys keep 28 bits per end; the workspace of that many points exceeds one MI355X anyway).
 * f4l_supervoxel_parallel = f4l_knn + f4l_normals + this (f4l_knn synchronises while it sizes its grid -- unless the stream is
 * being captured into a HIP graph or F4L_KNN_ASYNC is set: then the grid is sized on the device and the call only enqueues). */
size_t f4l_supervoxel_segment_device_workspace_bytes(int64_t n, int k);
int f4l_supervoxel_segment_device(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k,
                                  double resolution, const float *grid_bbox_host, int32_t *labels_out, int32_t *reps_out,
                                  int32_t *info_out, void *workspace, size_t workspace_bytes, void *stream);
size_t f4l_supervoxel_parallel_workspace_bytes(int64_t n, int k);
int f4l_supervoxel_parallel(const float *xyz, int64_t n, int k, double resolution, int32_t *labels_out, int32_t *reps_out,
                            int32_t *info_out, int32_t *knn_out, double *normals_out, void *workspace,
                            size_t workspace_bytes, void *stream);

/* The same partition in TWO calls sharing one workspace, for a caller who derives the resolution from the point spacing
 * (src/coarse_to_fine_matching_base.py:2668-2671: resolution = max(sqrt(3) * 10 * median spacing, voxel); the neighbour search
 * of cpp_core/supervoxel_segmentation/supervoxel.cpp:105-113 does not depend on it):
 *   f4l_partition_neighbours  kNN-k + PCA normals of the cloud in the search's own cell order ("position mode": neighbours named
 *       by position, lists transposed), left INSIDE the workspace; nn1_d2_out (device double [n], or NULL): every point's squared
 *       distance to its nearest other point, in that order -- good for the median of `_compute_median_resolution`
 *       (:2716-2754).  k <= 36 (the lane-per-query search); F4L_EUNSUPPORTED beyond: use f4l_knn_normals +
 *       f4l_supervoxel_segment_device.  Synchronises `stream` once like f4l_knn (not under graph capture / F4L_KNN_ASYNC).
 *   f4l_partition_segment     f4l_supervoxel_segment_device on what the first call left: the segmentation adopts the search's
 *       order as it stands (no ordering sort, no gather of the cloud, no transpose of the lists); outputs in the CALLER's order
 *       as always.  The workspace must not be touched between the two calls; n and k must be the same.
 * f4l_supervoxel_parallel runs the pair when neither knn_out nor normals_out is asked for. */
size_t f4l_partition_workspace_bytes(int64_t n, int k);
int f4l_partition_neighbours(const float *xyz, int64_t n, int k, double *nn1_d2_out, void *workspace, size_t workspace_bytes,
                             void *stream);
int f4l_partition_segment(int64_t n, int k, double resolution, const float *grid_bbox_host, int32_t *labels_out,
                          int32_t *reps_out, int32_t *info_out, void *workspace, size_t workspace_bytes, void *stream);

/* The reference's segmentation (supervoxel_segmentation.h:65-248) ON THE DEVICE, LABEL FOR LABEL: its sequential fusion
 * (:117-176: representatives in index order, ordered adjacency lists, the `break` at K) and its FIFO boundary exchange
 * (:186-237) computed as fixed points of parallel passes (csrc/supervoxel_exact.hip: what a representative does depends on the
 * round's start and on what LOWER-indexed ones did; every one is evaluated against an estimate of that until nothing changes --
 * the unique fixed point is the sequential result; 6-20 passes per lambda round).  xyz, normals, knn as for
 * f4l_supervoxel_segment_device (device arrays, the CALLER's order; knn rows in the order the search returned them: the
 * reference's queue order follows it).  labels_out int32 [n] (device), n_supervoxels_host (host), stats_host int32 [5] or NULL:
 * lambda rounds, fusion passes, exchange generations, exchange passes, the largest closure a representative had.  SYNCHRONISES `stream`.  F4L_EUNSUPPORTED when a
 * representative's closure or the lists outgrow the device buffers (k = 1, degenerate clouds): f4l_supervoxel then replays the
 * sequence on the host (f4l_supervoxel_segment_host) -- the same labels either way. */
size_t f4l_supervoxel_segment_exact_workspace_bytes(int64_t n, int k);
int f4l_supervoxel_segment_exact(const float *xyz, const double *normals, const int32_t *knn, int64_t n, int k, double resolution,
                                 int32_t *labels_out, int32_t *n_supervoxels_host, int32_t *stats_host, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* Host-only helper (no device work): the sequential segmentation stage on host arrays.  Exposed so the
 * Python shim can re-segment cached kNN/normals; same semantics as inside f4l_supervoxel. */
int f4l_supervoxel_segment_host(const float *xyz_host, const double *normals_host, const int32_t *knn_host,
                                int64_t n, int k, double resolution, int32_t *labels_host,
                                int32_t *n_supervoxels_host);

/* Host-only: write the partition text file `x y z r g b label` byte-for-byte as the reference does
 * (supervoxel.cpp:45-64 -> codelibrary/geometry/io/xyz_io.h:192-221); `load_partition`
 * (src/coarse_to_fine_matching_base.py:1257,1275) re-reads column 6. */
int f4l_write_partition_txt(const char *path, const float *xyz_host, const int32_t *labels_host, int64_t n,
                            int32_t n_supervoxels);

/* Host-only: the result files of a tile -- `np.savetxt(path, rows, delimiter=" ", fmt="%.6f")` as `save_process_dvf` calls it
 * (src/coarse_to_fine_matching_base.py:3477-3537) -- byte for byte: rows_host float32 [n][ncols], one row per line, values
 * separated by one blank, six decimals rounded as printf rounds the exact value.  (numpy formats a million rows of six in
 * 2.4 s, four to eight such files per tile; this writer takes 0.1 s.) */
int f4l_write_rows_txt(const char *path, const float *rows_host, int64_t n, int ncols);

/* Sort-by-label -> CSR (replaces the O(K*N) mask loop of prepare_pts2spt_dict,
 * src/coarse_to_fine_matching_base.py:1327-1332).  labels int32 [n]; order_out int32 [n] = point ids grouped by label
 * (stable); off_out int64 [K+1].  A label outside [0, K) (an "unlabelled" -1, a label beyond the caller's count) belongs to no
 * patch: those point ids follow the last patch, order_out[off_out[K] ..), in ascending order. */
size_t f4l_labels_to_csr_workspace_bytes(int64_t n, int64_t K);
int f4l_labels_to_csr(const int32_t *labels, int64_t n, int64_t K, int32_t *order_out, int64_t *off_out,
                      void *workspace, size_t workspace_bytes, void *stream);
/* The same for a cloud whose m rows take their labels from ANOTHER cloud's rows: row i belongs to patch labels[via[i]]
 * (`labels[nn]` followed by f4l_labels_to_csr without materialising the gathered labels: how the second epoch's points join the
 * patch of their nearest first-epoch point, via = f4l_epoch_join's tgt_to_src_out).  A via outside [0, n_labels) is "no patch".
 * Workspace: f4l_labels_to_csr_workspace_bytes(m, K). */
int f4l_labels_to_csr_via(const int32_t *labels, int64_t n_labels, const int32_t *via, int64_t m, int64_t K, int32_t *order_out,
                          int64_t *off_out, void *workspace, size_t workspace_bytes, void *stream);

/* The two searches the path runs over the SECOND epoch of a tile, with one binning of it: tgt_nn1_d2_out double [m] (nullable) =
 * every target point's squared distance to its nearest other target point -- what `_compute_median_resolution`
 * (src/coarse_to_fine_matching_base.py:2716-2754) takes the median of; tgt_to_src_out int32 [m] = index of its nearest SOURCE
 * point (ties by index) -- f4l_knn(tgt, 2) and f4l_nn_query(src, n, tgt, m, 1) in one call: same values, the target cloud
 * sorted once instead of twice.  Synchronises `stream` (grid sizing).  Non-finite coordinates: F4L_EINVAL. */
size_t f4l_epoch_join_workspace_bytes(int64_t n, int64_t m);
int f4l_epoch_join(const float *src, int64_t n, const float *tgt, int64_t m, double *tgt_nn1_d2_out, int32_t *tgt_to_src_out,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Median of n doubles values[i * stride] as numpy.median computes it (the middle element, or the mean of the middle pair) --
 * the last step of `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754: median of the nearest-
 * neighbour distances).  The values must not be NaN.  median_out: DEVICE double [1].  Device radix sort; no synchronisation. */
size_t f4l_median_f64_workspace_bytes(int64_t n);
int f4l_median_f64(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace, size_t workspace_bytes,
                   void *stream);
/* numpy.median(numpy.sqrt(values)) for values >= 0 (squared distances in, the median distance out) without a pass for the roots. */
int f4l_median_sqrt_f64(const double *values, int64_t n, int64_t stride, double *median_out, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The tiling front end of both entry scripts (main_piecewise_icp.py:60-83, main_fusion.py:112-125 -> src/functions.py:147-177 ->
 * the SWIG module cpp_core/pcd_tiling/pcd_tiling.i), `tile_point_clouds` of cpp_core/pcd_tiling/pcd_tiling.h:3-12, body
 * cpp_core/pcd_tiling/pcd_tiling.cpp:709-871 -- the same ten arguments in the same order: crop both epochs to the overlap of their
 * bounding boxes (:73-116), optionally thin them with a voxel grid (pcl::VoxelGrid, :118-227; voxel_grid_filter_size == 0: the median
 * nearest-neighbour spacing of the smaller cloud, :37-54), halve the box along the longer side of the projection plane until both
 * halves hold fewer than max_points_per_tile points (:231-655), and write every leaf as
 *   <save_dir>/non_overlap/{source,target}_tile_<i>.ply and <save_dir>/overlap/{source,target}_tile_<i>_overlap.ply
 * (binary little-endian PLY, float x y z [+ uchar red green blue]; the overlap twin: the leaf's box grown by 20 m in the projection
 * plane -- hard coded in the reference, which accepts and ignores min_points_per_tile and overlap_tiles: so does this).
 * projection_direction: -1 (the axis whose face of the overlap box is largest), 0, 1 or 2.
 * Both clouds live on the device from the read to the leaves: boxes, crops (stable compaction), the voxel grid with its colour
 * averages, the spacing estimate and the counts that steer the recursion are kernels; the host walks the tree of boxes and writes
 * the files.  A FILE-LEVEL entry: paths in, files out; it allocates its own device memory and synchronises `stream`.
 * *n_tiles_host (HOST int, nullable): the tiles written per epoch; -2 when an input file does not exist (the reference prints and
 * returns false there, :735-738: the return value is F4L_OK).  F4L_EUNSUPPORTED: more than max_points_per_tile coincident points
 * (the reference recurses until its stack overflows), or list properties on the vertex element.  PCL's filter semantics are restated
 * from its documentation [parity unpinned]; file for file the output equals oracle/pcd_tiling_ref.py (tests/test_pcd_tiling.py). */
int f4l_tile_point_clouds(const char *first_ply, const char *second_ply, int max_points_per_tile, int min_points_per_tile,
                          int voxel_grid_flag, float voxel_grid_filter_size, float overlap_tiles, int projection_direction,
                          const char *save_dir, int verbose, int32_t *n_tiles_host, void *stream);
/* `resave_point_cloud` (pcd_tiling.h:15-17, pcd_tiling.cpp:662-707): both files re-written as binary PLY (host only).
 * *ok_host: 1 done, 0 an input file does not exist. */
int f4l_resave_point_cloud(const char *first_ply, const char *second_ply, int verbose, int32_t *ok_host);

/* Gather rows: out[i] = pts[order[i]] (float32 [n][3]); builds patch-contiguous clouds from a CSR order. */
int f4l_gather_points(const float *pts, const int32_t *order, int64_t n, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* F4L_H_ */
