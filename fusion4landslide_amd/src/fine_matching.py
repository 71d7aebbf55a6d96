"""The 3D part of `fine_matching_with_different_types` (src/coarse_to_fine_matching_base.py:3236-3436), the reference's
per-patch-match Python loop, for ALL patch matches of a tile at once.  One call replaces, per match:

    torch.isin gather of the mutual point matches            :3259-3274   f4l_mutual_correspondences
    (remove_low_quality_patch_matches) two n x n cdist         :3304-3320   f4l_rigidity_check
    refine_local_rigid_correspondences (weighted Kabsch)       :3341        |
    tensor2pcd x 2, icp_registration on the MUTUAL points      :3352-3360   |  f4l_patch_loop (one launch)
    transform applied to ALL points of the source patch        :3371-3374   |
    (output_tgt2src) inverse applied to the target patch       :3393-3397   f4l_apply_transform(inverse)
    assign_all_src / assign_then_nn sparse rows                :3410-3434   f4l_apply_transform / f4l_nn_refine

The image-matching halves of the reference's loop (`fine_matching_only_2d`, `fine_matching_fusion`, `weighting_svd`) need
the 2D correspondences its out-of-scope image pipeline produces; this mirror is the `fine_matching_only_3d` path.

Patch matches are CSR: match i pairs the source points `src_ids[src_off[i]:src_off[i+1]]` with the target points
`tgt_ids[tgt_off[i]:tgt_off[i+1]]` (`spt_corres_src[i]` / `spt_corres_tgt[i]`; ids ascending inside a target patch).
"""
from .. import engine


def fine_matching_3d(src_pts, tgt_pts, src_ids, src_off, tgt_ids, tgt_off, corr_tgt, *, num_min_fine_match=3,
                     icp_threshold=0.1, remove_low_quality_patch_matches=False, num_min_matches_for_quality_check=10,
                     thres_dist_diff=0.05, thres_inlier_ratio=0.5, assign_type="assign_all_src", output_tgt2src=False,
                     median_max_resolution=0.0, icp_type="point2point"):
    """Returns a dict:
      dense        (m, 6) float32 [s, T s] for every point of every registered match's source patch, in match order (:3408)
      sparse       (k, 6) float32: assign_all_src [mutual s, T mutual s] (:3413-3414); assign_then_nn the rows of
                   refine_dvfs_with_threshold, each match's block twice in a row like the reference appends it (:3427-3434)
      tgt2src      (l, 6) float32 [T^-1 q, q] over the target patches (:3393-3397) when output_tgt2src, else None
      mask_useful  (P,) bool   False where the quality check dropped the match (:3322-3325)
      mask_global  (P,) bool   False where the match had fewer than num_min_fine_match mutual pairs (:3436)
      metric       (P, 2) float64 [ratio_inlier, dist_mean] ([0, 0] below num_min_matches_for_quality_check, :3332) or None
      T, fitness, rmse, iters  per match (iters == -1: not registered)
    """
    import torch
    if assign_type not in ("assign_all_src", "assign_then_nn"):
        raise NotImplementedError(assign_type)
    dev = src_pts.device
    P = src_off.shape[0] - 1
    src_ids, tgt_ids = src_ids.to(torch.int64), tgt_ids.to(torch.int64)
    # mutual point matches of every patch match
    mask, count = engine.mutual_correspondences(src_ids, src_off, tgt_ids, tgt_off, corr_tgt)
    n_src_rows = src_off[1:] - src_off[:-1]
    pid_rows = torch.repeat_interleave(torch.arange(P, device=dev), n_src_rows)

    def pairs(keep_rows):
        s = src_ids[keep_rows]
        return src_pts[s].contiguous(), tgt_pts[corr_tgt[s]].contiguous()

    mask_useful = torch.ones(P, dtype=torch.bool, device=dev)
    metric = None
    if remove_low_quality_patch_matches:
        cs, ct = pairs(mask)
        coff = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        coff[1:] = torch.cumsum(count, 0)
        dist_mean, ratio_inlier = engine.rigidity_check(cs, ct, coff, thres_dist_diff)
        checked = count >= num_min_matches_for_quality_check
        bad = checked & ((ratio_inlier <= thres_inlier_ratio) | (dist_mean >= thres_dist_diff))
        mask_useful = ~bad
        metric = torch.stack([torch.where(checked, ratio_inlier, torch.zeros_like(ratio_inlier)),
                              torch.where(checked, dist_mean, torch.zeros_like(dist_mean))], dim=1)
        mask = mask & mask_useful[pid_rows]  # a dropped match takes no further part (`continue`, :3325)
        count = torch.where(mask_useful, count, torch.zeros_like(count))
    mask_global = ~(mask_useful & (count < num_min_fine_match))
    cs, ct = pairs(mask)
    coff = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    coff[1:] = torch.cumsum(count, 0)
    # Kabsch -> ICP on the mutual points -> rows of all source-patch points, one launch; matches below the minimum are skipped
    rows_src = src_pts[src_ids].contiguous()
    skip_below = max(int(num_min_fine_match), 1)  # (a dropped match has no pairs left: it must not start from the identity)
    out = engine.patch_loop(cs, coff, ct, coff, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=icp_threshold, max_iter=30,
                            rel_fitness=1e-6, rel_rmse=1e-6, icp_type=icp_type, rows_src=rows_src, rows_off=src_off,
                            min_corr=skip_below)
    done = out["iters"] >= 0
    dense = out["rows"][done[pid_rows]]
    res = dict(dense=dense, mask_useful=mask_useful, mask_global=mask_global, metric=metric, T=out["T"], fitness=out["fitness"],
               rmse=out["rmse"], iters=out["iters"], tgt2src=None)
    if output_tgt2src:
        n_tgt_rows = tgt_off[1:] - tgt_off[:-1]
        pid_t = torch.repeat_interleave(torch.arange(P, device=dev), n_tgt_rows)
        res["tgt2src"] = engine.apply_transform(tgt_pts[tgt_ids].contiguous(), tgt_off, out["T"], inverse=True)[done[pid_t]]
    if assign_type == "assign_all_src":
        pid_c = torch.repeat_interleave(torch.arange(P, device=dev), count)
        res["sparse"] = engine.apply_transform(cs, coff, out["T"])[done[pid_c]]
    else:
        thr = out["rmse"] * 2.0  # :3420-3424
        thr = torch.where(torch.isfinite(thr), thr, torch.full_like(thr, float(median_max_resolution)))
        thr = torch.clamp(thr, min=float(median_max_resolution))
        nn, rows = engine.nn_refine(rows_src, src_off, tgt_pts[tgt_ids].contiguous(), tgt_off, out["T"], thr)
        keep = (nn >= 0) & done[pid_rows]
        rows, pid_k = rows[keep], pid_rows[keep]
        # the reference appends every match's block twice (:3428 and :3434)
        order = torch.argsort(torch.cat([2 * pid_k, 2 * pid_k + 1]), stable=True)
        res["sparse"] = torch.cat([rows, rows])[order]
    return res
