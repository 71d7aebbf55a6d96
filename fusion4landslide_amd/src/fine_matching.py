"""`fine_matching_with_different_types` (src/coarse_to_fine_matching_base.py:3236-3436), the reference's per-patch-match Python
loop, for ALL patch matches of a tile at once.  One call replaces, per match:

    torch.isin gather of the mutual point matches, from the 3D       :3259-3261   f4l_mutual_correspondences
      matching and / or from the lifted 2D matching                    :3262-3267   f4l_mutual_correspondences (second set)
    their concatenation, 3D first (`fine_matching_fusion`)             :3269-3276
    (weighting_svd) the weight vector of the Kabsch fit                :3282-3296
    (remove_low_quality_patch_matches) two n x n cdist                 :3304-3334   f4l_rigidity_check
    refine_local_rigid_correspondences (weighted Kabsch)               :3341        |
    tensor2pcd x 2, icp_registration on the MUTUAL points              :3352-3360   |  f4l_patch_loop (one launch)
    transform applied to ALL points of the source patch                :3371-3374   |
    (output_tgt2src) inverse applied to the target patch               :3393-3397   f4l_apply_transform(inverse)
    assign_all_src / assign_then_nn sparse rows                        :3410-3434   f4l_apply_transform / f4l_nn_refine

The image pipeline that PRODUCES the 2D matches is out of scope; their result, `corres_3d_from_2d_idx` --
for every source point the target point its pixel match lifts to, -1 for none (:1670-1675) --, is an input here, like the 3D
matches `corres_3d_voxel_from_3d_idx` (:2879-2889).

Patch matches are CSR: match i pairs the source points `src_ids[src_off[i]:src_off[i+1]]` with the target points
`tgt_ids[tgt_off[i]:tgt_off[i+1]]` (`spt_corres_src[i]` / `spt_corres_tgt[i]`; ids ascending inside a target patch).
"""
from .. import engine


def fine_matching_3d(src_pts, tgt_pts, src_ids, src_off, tgt_ids, tgt_off, corr_tgt, **kw):
    """prepare -> ONE f4l_patch_loop launch -> finish; see :func:`fine_matching_prepare` for the arguments and the result.  The two
    halves are separate so that the launches of SEVERAL tiles can be one (main_fusion's tile loop: engine.patch_loop_tiles)."""
    st = fine_matching_prepare(src_pts, tgt_pts, src_ids, src_off, tgt_ids, tgt_off, corr_tgt, **kw)
    a = st["loop_args"]
    out = engine.patch_loop(a["src"], a["src_off"], a["tgt"], a["tgt_off"], a["corr_src"], a["corr_ref"], a["corr_off"], a["corr_weights"],
                            0.0, 1e-6, rows_src=a["rows_src"], rows_off=a["rows_off"], **st["loop_kw"])
    return fine_matching_finish(st, out)


def fine_matching_prepare(src_pts, tgt_pts, src_ids, src_off, tgt_ids, tgt_off, corr_tgt, *, corr_tgt_2d=None, matching="only_3d",
                          weighting_svd=False, num_min_fine_match=3, icp_threshold=0.1, remove_low_quality_patch_matches=False,
                          num_min_matches_for_quality_check=10, thres_dist_diff=0.05, thres_inlier_ratio=0.5, assign_type="assign_all_src",
                          output_tgt2src=False, median_max_resolution=0.0, icp_type="point2point", init_round_f32=True,
                          rigidity_precision="f64"):
    """corr_tgt (n_src_points,) int64: `corres_3d_voxel_from_3d_idx[:, 1]`, the target point matched to each source point, -1 for
    none; corr_tgt_2d: `corres_3d_from_2d_idx[:, 1]` likewise (needed for matching = "only_2d" / "fusion").
    matching: "only_3d" | "only_2d" | "fusion" -- `method.fine_matching_only_3d / _only_2d / _fusion` (:3257-3276); in "fusion" a
    match's pairs are the 3D ones followed by the 2D ones (a source point matched by both appears twice, as in the reference).
    weighting_svd (:3282-3296, "fusion" only -- the reference's other two modes leave one of the two counts undefined): with n3 / n2
    pairs from the 3D / 2D matching, weight n3 / (n3 + n2) for the first n3 pairs, then -- the reference's own overwrite, indexed
    with n2 where n3 was meant -- 0.01 from pair n2 on, 1 in between; a match that passes the quality check is fitted without
    weights (:3329).
    init_round_f32: ICP starts from the float32 values of the Kabsch transform (the reference's float32 4 x 4, :3360).
    rigidity_precision: "f64" | "f32", the pair arithmetic of the quality check (engine.rigidity_check).

    Returns the state :func:`fine_matching_finish` completes once the per-patch loop has run on `loop_args` (the tile dict of
    engine.merge_tiles / the arguments of engine.patch_loop) with `loop_kw`.  fine_matching_finish returns a dict:
      dense        (m, 6) float32 [s, T s] for every point of every registered match's source patch, in match order (:3408)
      sparse       (k, 6) float32: assign_all_src [mutual s, T mutual s] (:3413-3414); assign_then_nn the rows of
                   refine_dvfs_with_threshold, each match's block twice in a row like the reference appends it (:3427-3434)
      tgt2src      (l, 6) float32 [T^-1 q, q] over the target patches (:3393-3397) when output_tgt2src, else None
      mask_useful  (P,) bool   False where the quality check dropped the match (:3322-3325)
      mask_global  (P,) bool   False where the match had fewer than num_min_fine_match mutual pairs (:3436)
      metric       (P, 2) float64 [ratio_inlier, dist_mean] ([0, 0] below num_min_matches_for_quality_check, :3332) or None
      T, fitness, rmse, iters  per match (iters == -1: not registered);  n_pairs (P, 2) int64: pairs from the 3D / the 2D matching
    """
    import torch
    if assign_type not in ("assign_all_src", "assign_then_nn"):
        raise NotImplementedError(assign_type)
    if matching not in ("only_3d", "only_2d", "fusion"):
        raise NotImplementedError(matching)  # :3275-3276
    if matching != "only_3d" and corr_tgt_2d is None:
        raise ValueError("matching = %r needs corr_tgt_2d, the target point every source point's 2D match lifts to" % matching)
    if weighting_svd and matching != "fusion":
        raise ValueError("weighting_svd needs the pairs of both matchings (fine_matching_fusion; :3284-3286)")
    dev = src_pts.device
    P = src_off.shape[0] - 1
    src_ids, tgt_ids = src_ids.to(torch.int64), tgt_ids.to(torch.int64)
    n_src_rows = src_off[1:] - src_off[:-1]
    pid_rows = torch.repeat_interleave(torch.arange(P, device=dev), n_src_rows)
    zero = torch.zeros(P, dtype=torch.int64, device=dev)
    # mutual point matches of every patch match, per source of matches
    mask3, n3 = engine.mutual_correspondences(src_ids, src_off, tgt_ids, tgt_off, corr_tgt) if matching != "only_2d" else (None, zero)
    mask2, n2 = engine.mutual_correspondences(src_ids, src_off, tgt_ids, tgt_off, corr_tgt_2d) if matching != "only_3d" else (None, zero)

    def pair_list(keep3, keep2):
        """The matches' pair lists, 3D pairs before 2D pairs inside a match (:3273): (source rows, target ids, match of every pair)."""
        rows, tids, pids = [], [], []
        for keep, corr in ((keep3, corr_tgt), (keep2, corr_tgt_2d)):
            if keep is None:
                continue
            r = torch.nonzero(keep, as_tuple=True)[0]
            rows.append(src_ids[r])
            tids.append(corr[src_ids[r]])
            pids.append(pid_rows[r])
        if len(rows) == 1:  # (one matching: the pairs already stand in match order)
            return rows[0], tids[0], pids[0]
        rows, tids, pids = torch.cat(rows), torch.cat(tids), torch.cat(pids)
        order = torch.argsort(pids, stable=True)  # (stable: a match's 3D pairs stay ahead of its 2D pairs, each in patch order)
        return rows[order], tids[order], pids[order]

    def offsets(count):
        off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(count, 0)
        return off

    count = n3 + n2
    mask_useful = torch.ones(P, dtype=torch.bool, device=dev)
    metric = None
    passed_check = torch.zeros(P, dtype=torch.bool, device=dev)
    if remove_low_quality_patch_matches:
        rows, tids, _ = pair_list(mask3, mask2)
        dist_mean, ratio_inlier = engine.rigidity_check(src_pts[rows].contiguous(), tgt_pts[tids].contiguous(), offsets(count), thres_dist_diff,
                                                        precision=rigidity_precision)
        checked = count >= num_min_matches_for_quality_check
        bad = checked & ((ratio_inlier <= thres_inlier_ratio) | (dist_mean >= thres_dist_diff))
        mask_useful = ~bad
        passed_check = checked & ~bad
        metric = torch.stack([torch.where(checked, ratio_inlier, torch.zeros_like(ratio_inlier)),
                              torch.where(checked, dist_mean, torch.zeros_like(dist_mean))], dim=1)
        keep_row = mask_useful[pid_rows]  # a dropped match takes no further part (`continue`, :3325)
        mask3 = None if mask3 is None else mask3 & keep_row
        mask2 = None if mask2 is None else mask2 & keep_row
        n3 = torch.where(mask_useful, n3, zero)
        n2 = torch.where(mask_useful, n2, zero)
        count = n3 + n2
    mask_global = ~(mask_useful & (count < num_min_fine_match))
    rows, tids, pids = pair_list(mask3, mask2)
    coff = offsets(count)
    cs, ct = src_pts[rows].contiguous(), tgt_pts[tids].contiguous()
    weights = None
    if weighting_svd:
        k = torch.arange(rows.shape[0], device=dev) - coff[pids]  # place of a pair inside its match's list
        wv = (n3.to(torch.float64) / torch.clamp(count, min=1).to(torch.float64)).to(torch.float32)[pids]  # (a Python float stored into a float32 tensor)
        weights = torch.where(k < n3[pids], wv, torch.ones_like(wv))
        weights = torch.where(k >= n2[pids], torch.full_like(wv, 0.01), weights)  # :3290-3294, the overwrite included
        weights = torch.where(passed_check[pids], torch.ones_like(wv), weights)    # :3329 `weight_vector = None`
    # Kabsch -> ICP on the mutual points -> rows of all source-patch points, one launch; matches below the minimum are skipped
    rows_src = src_pts[src_ids].contiguous()
    skip_below = max(int(num_min_fine_match), 1)  # (a dropped match has no pairs left: it must not start from the identity)
    loop_args = dict(src=cs, src_off=coff, tgt=ct, tgt_off=coff, corr_src=cs, corr_ref=ct, corr_off=coff, corr_weights=weights,
                     rows_src=rows_src, rows_off=src_off)
    loop_kw = dict(max_corr_dist=icp_threshold, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, icp_type=icp_type, min_corr=skip_below,
                   init_round_f32=init_round_f32)
    return dict(loop_args=loop_args, loop_kw=loop_kw, P=P, dev=dev, pid_rows=pid_rows, pids=pids, mask_useful=mask_useful,
                mask_global=mask_global, metric=metric, n3=n3, n2=n2, cs=cs, coff=coff, rows_src=rows_src, src_off=src_off,
                tgt_pts=tgt_pts, tgt_ids=tgt_ids, tgt_off=tgt_off, output_tgt2src=output_tgt2src, assign_type=assign_type,
                median_max_resolution=median_max_resolution)


def fine_matching_finish(st, out):
    """What follows the per-patch loop (:3371-3436) for one tile: `out` = the loop's result for this tile's patch matches (T, fitness,
    rmse, iters, rows -- engine.patch_loop, or this tile's slice of engine.patch_loop_tiles)."""
    import torch
    P, dev, pid_rows, pids = st["P"], st["dev"], st["pid_rows"], st["pids"]
    tgt_pts, tgt_ids, tgt_off = st["tgt_pts"], st["tgt_ids"], st["tgt_off"]
    done = out["iters"] >= 0
    dense = out["rows"][done[pid_rows]]
    res = dict(dense=dense, mask_useful=st["mask_useful"], mask_global=st["mask_global"], metric=st["metric"], T=out["T"],
               fitness=out["fitness"], rmse=out["rmse"], iters=out["iters"], tgt2src=None, n_pairs=torch.stack([st["n3"], st["n2"]], dim=1))
    if st["output_tgt2src"]:
        n_tgt_rows = tgt_off[1:] - tgt_off[:-1]
        pid_t = torch.repeat_interleave(torch.arange(P, device=dev), n_tgt_rows)
        res["tgt2src"] = engine.apply_transform(tgt_pts[tgt_ids].contiguous(), tgt_off, out["T"], inverse=True)[done[pid_t]]
    if st["assign_type"] == "assign_all_src":
        res["sparse"] = engine.apply_transform(st["cs"], st["coff"], out["T"])[done[pids]]
    else:
        thr = out["rmse"] * 2.0  # :3420-3424
        thr = torch.where(torch.isfinite(thr), thr, torch.full_like(thr, float(st["median_max_resolution"])))
        thr = torch.clamp(thr, min=float(st["median_max_resolution"]))
        nn, rws = engine.nn_refine(st["rows_src"], st["src_off"], tgt_pts[tgt_ids].contiguous(), tgt_off, out["T"], thr)
        keep = (nn >= 0) & done[pid_rows]
        rws, pid_k = rws[keep], pid_rows[keep]
        # the reference appends every match's block twice (:3428 and :3434): row l of a match with c rows, o rows before it, goes to
        # 2 o + l and to 2 o + c + l
        c = torch.bincount(pid_k, minlength=P)
        o = torch.cumsum(c, 0) - c
        pos = torch.arange(rws.shape[0], device=dev) + o[pid_k]  # (= 2 o + l, with l = index - o)
        sparse = torch.empty((2 * rws.shape[0], rws.shape[1]), dtype=rws.dtype, device=dev)
        sparse[pos] = rws
        sparse[pos + c[pid_k]] = rws
        res["sparse"] = sparse
    return res
