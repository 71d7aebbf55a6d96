"""Mirror of the part of the reference's src/rgb_guided.py that sits on the per-patch rigid-fit path (the rest of that file is
image lifting and matching: out of scope): its two helpers, the patch assembly that ends `implement_segmentation`, and the
per-patch loop `local_rigid_refinement` -- the RGB-guided home of the SVD -> ICP loop, batched.

    segment_patches_from_labels(...)                           src/rgb_guided.py:935-979 (the tail of implement_segmentation)
    local_rigid_refinement_batched(...)                        src/rgb_guided.py:981-1062

    weighted_procrustes(...)                                   src/rgb_guided.py:25-96  -- a verbatim duplicate of
                                                               scripts/weighted_svd.py:58-129: the same function here
    refine_local_rigid_correspondences(corr_neigh_2, 'SVD')    src/rgb_guided.py:99-125 -- NOT the function of the same name in
                                                               scripts/weighted_svd.py: it prunes at 2.5 x the median residual
                                                               and returns four values
"""
from .. import engine
from ..scripts.weighted_svd import weighted_procrustes  # noqa: F401  (src/rgb_guided.py:25-96 duplicates it)


def refine_local_rigid_correspondences(corr_neigh_2, refine_type='SVD'):
    """src/rgb_guided.py:99-125: Kabsch (eps = 1e-6, no weights) of the (n, 6) rows [src, tgt]; residual norms; keep the rows
    below 2.5 x the median residual (torch.median: the LOWER median); `mask_2` says whether at least 70 % survived.
    Returns (pruned (n', 6), transform (4, 4) float32 on the GPU, mask (n,) bool, mask_2 0-d bool tensor)."""
    import torch
    if refine_type != 'SVD':
        # the reference's 'RANSAC' branch returns a single value where its caller unpacks four (:126-133): unusable there too
        raise NotImplementedError("only refine_type='SVD' is implemented")
    rot, tra = weighted_procrustes(corr_neigh_2[:, :3], corr_neigh_2[:, 3:6], weights=None, weight_thresh=0.0, eps=1e-6,
                                   return_transform=False)
    off = torch.tensor([0, corr_neigh_2.shape[0]], dtype=torch.int64, device=corr_neigh_2.device)
    res = engine.kabsch_residuals(corr_neigh_2[:, :3], corr_neigh_2[:, 3:6], off, rot.unsqueeze(0), tra.unsqueeze(0))
    res = res.to(corr_neigh_2.dtype)  # the reference's residuals are in the clouds' dtype (float32 in practice)
    mask = res < 2.5 * torch.median(res)
    mask_2 = torch.sum(mask) / res.shape[0] >= 0.70
    T = torch.eye(4, device=corr_neigh_2.device)
    T[:3, :3] = rot
    T[:3, 3] = tra
    return corr_neigh_2[mask, :], T, mask, mask_2


def refine_local_rigid_correspondences_batched(corr, off):
    """The same for all patch matches of a tile at once: corr (n, 6) rows grouped by `off` (P + 1,).  Returns (keep mask
    (n,) bool, T (P, 4, 4) float64, mask_2 (P,) bool)."""
    import torch
    src, ref = corr[:, :3].contiguous(), corr[:, 3:6].contiguous()
    R, t = engine.kabsch_batched(src, ref, off, None, 0.0, 1e-6)
    res = engine.kabsch_residuals(src, ref, off, R, t).to(corr.dtype)
    P = off.shape[0] - 1
    cnt = off[1:] - off[:-1]
    pid = torch.repeat_interleave(torch.arange(P, device=corr.device), cnt)
    # lower median per patch: sort the residuals inside every patch, take element (n - 1) // 2
    by_res = torch.argsort(res, stable=True)
    order = by_res[torch.argsort(pid[by_res], stable=True)]  # rows grouped by patch, ascending residual inside a patch
    med = torch.zeros(P, dtype=res.dtype, device=corr.device)
    has = cnt > 0
    med[has] = res[order][(off[:-1] + (cnt - 1).clamp(min=0) // 2)[has]]
    keep = res < 2.5 * med[pid]
    kept = torch.zeros(P, dtype=torch.int64, device=corr.device).index_add_(0, pid, keep.to(torch.int64))
    mask_2 = kept.to(torch.float32) / cnt.clamp(min=1).to(torch.float32) >= 0.70
    T = torch.eye(4, dtype=torch.float64, device=corr.device).unsqueeze(0).repeat(P, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return keep, T, mask_2


def segment_patches_from_labels(segment_id_src_pts_input, idx_valid_src, idx_valid_tgt, corres_3d, corres_3d_magnitude,
                                min_count=10):
    """The patch assembly at the end of `implement_segmentation` (src/rgb_guided.py:935-979), without its Python loop over the
    segments: `segment_id_src_pts_input` (N,) the label of every input source point (what computeSupervoxel returned, :883-888),
    `idx_valid_src / idx_valid_tgt` (m,) the source / target point of every 3D correspondence, `corres_3d` (m, 6),
    `corres_3d_magnitude` (m, 1).  A segment stays when MORE than `min_count` of the valid points carry it and its id is not -1
    (:949-950); the points of the other segments leave the correspondence arrays (:964-977).

    Returns dict(segment_patches: list of id tensors, one per kept segment in ascending segment id, each = idx_valid_src of the
    segment's points in ascending row order (:954-956); segment_off (P + 1,) and segment_ids (cat of the list) -- the same as
    CSR, what `local_rigid_refinement_batched` takes --; idx_valid_src_refine, idx_valid_tgt_refine, corres_3d_refine; and
    corres_3d_magnitude_refine = corres_3d_magnitude UNFILTERED: the reference's statement is broken over two lines
    (`... = self.data_output.corres_3d_magnitude` / `[mask_pts_valid]`, :976-977), so its mask never applies; reproduced)."""
    import torch
    dev = corres_3d.device
    lab_all = torch.as_tensor(segment_id_src_pts_input, device=dev).reshape(-1).to(torch.int64)
    ivs = torch.as_tensor(idx_valid_src, device=dev).reshape(-1).to(torch.int64)
    ivt = torch.as_tensor(idx_valid_tgt, device=dev).reshape(-1)
    lab = lab_all[ivs]                                   # :938 segment ids of the valid points
    order = torch.argsort(lab, stable=True)              # rows grouped by segment id, ascending row inside a segment
    uniq, cnt = torch.unique_consecutive(lab[order], return_counts=True)   # np.unique order (:941)
    valid_seg = (cnt > min_count) & (uniq != -1)
    row_ok_sorted = torch.repeat_interleave(valid_seg, cnt)
    rows_kept_sorted = order[row_ok_sorted]              # rows of the kept segments, segment by segment
    seg_cnt = cnt[valid_seg]
    off = torch.zeros(seg_cnt.shape[0] + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(seg_cnt, 0)
    seg_ids = ivs[rows_kept_sorted]
    mask_pts_valid = torch.zeros(ivs.shape[0], dtype=torch.bool, device=dev)
    mask_pts_valid[rows_kept_sorted] = True
    bounds = off.tolist()
    return dict(segment_patches=[seg_ids[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1)],
                segment_ids=seg_ids, segment_off=off, mask_pts_valid=mask_pts_valid,
                idx_valid_src_refine=ivs[mask_pts_valid], idx_valid_tgt_refine=ivt[mask_pts_valid],
                corres_3d_refine=corres_3d[mask_pts_valid, :], corres_3d_magnitude_refine=corres_3d_magnitude)


def local_rigid_refinement_batched(corres_3d_refine, idx_valid_src_refine, segment_patches, icp_thres, icp_refine=True,
                                   idx_valid_tgt_refine=None, corres_3d_magnitude_refine=None, segment_off=None, search="f64"):
    """`local_rigid_refinement` (src/rgb_guided.py:981-1062) for ALL segment patches of a tile in three launches instead of its
    Python loop with two device crossings per patch.  Per patch the reference
      1. collects the correspondence rows whose source point is in the patch, in the order of the patch's ids (:987-993),
      2. fits them rigidly (Kabsch, eps = 1e-6), prunes at 2.5 x the LOWER median residual, notes whether 70 % survive
         (`mask_robust`, computed and not used, :994-996 / 99-125) and keeps the surviving ROW INDICES (`mask_valid_local`, :1007),
      3. with `icp_refine` and at least one row: point-to-point ICP (threshold `icp_thres`, criteria 1e-6 / 1e-6 / 30) of ALL the
         patch's rows -- sources against targets, NOT the pruned ones -- from the float32 Kabsch transform (:1009-1020), and rows
         [src, T_icp src] with the float32 values of T_icp for all of them (:1022-1047),
    and afterwards filters the correspondence arrays by the kept rows and stacks the ICP rows (:1050-1061).

    Here: step 1 by one sort + searchsorted; steps 2 and the Kabsch of 3 by `refine_local_rigid_correspondences_batched`; step 3
    by ONE `f4l_patch_loop` launch (Kabsch from all rows -> float32-rounded start -> ICP on (rows' sources, rows' targets) ->
    displacement rows).  `segment_patches`: list of id tensors (as `segment_patches_from_labels` / the reference build it), or the
    concatenated ids with `segment_off`.  A patch none of whose ids has a correspondence row contributes nothing (the reference
    cannot meet one: torch.median of no residuals raises).

    Returns dict(mask_valid_local, idx_valid_src_refine, idx_valid_tgt_refine, corres_3d_refine, corres_3d_magnitude_refine --
    the filtered arrays of :1050-1056 (None where the input was not given) --, corres_3d_refine_apply_icp (n_rows, 6),
    corres_3d_magnitude_refine_apply_icp (n_rows, 1) when `icp_refine`, and per patch: mask_robust (P,), T_init (P, 4, 4)
    float32, T_icp (P, 4, 4) float64, fitness, inlier_rmse, iters, patch_off (P + 1,) into the rows)."""
    import torch
    dev = corres_3d_refine.device
    ivs = torch.as_tensor(idx_valid_src_refine, device=dev).reshape(-1).to(torch.int64)
    if segment_off is None:
        sizes = [int(p.numel()) for p in segment_patches]
        vals = (torch.cat([torch.as_tensor(p, device=dev).reshape(-1).to(torch.int64) for p in segment_patches])
                if sizes else torch.zeros(0, dtype=torch.int64, device=dev))
        voff = torch.zeros(len(sizes) + 1, dtype=torch.int64, device=dev)
        voff[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.int64, device=dev), 0)
    else:
        vals = torch.as_tensor(segment_patches, device=dev).reshape(-1).to(torch.int64)
        voff = torch.as_tensor(segment_off, device=dev).to(torch.int64)
    P = voff.shape[0] - 1
    # 1. rows per patch: for every id of the patch, in order, the rows whose source point it is (ascending): `torch.where(
    #    idx_valid_src_refine == value)[0]` for value in patch_i, hstack-ed (:990-991)
    sorted_ids, perm = torch.sort(ivs, stable=True)
    lo, hi = torch.searchsorted(sorted_ids, vals, right=False), torch.searchsorted(sorted_ids, vals, right=True)
    per_val = hi - lo
    nrow = int(per_val.sum())
    val_of_row = torch.repeat_interleave(torch.arange(vals.shape[0], device=dev), per_val)
    first = torch.cumsum(per_val, 0) - per_val
    idx_all = perm[lo[val_of_row] + (torch.arange(nrow, device=dev) - first[val_of_row])]
    pid_of_val = torch.repeat_interleave(torch.arange(P, device=dev), voff[1:] - voff[:-1])
    rows_per_patch = torch.zeros(P, dtype=torch.int64, device=dev).index_add_(0, pid_of_val, per_val)
    off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(rows_per_patch, 0)
    corr = corres_3d_refine[idx_all, :].to(torch.float32).contiguous()
    # 2. Kabsch of all rows, residuals, 2.5 x lower-median prune
    keep, T_kabsch, mask_robust = refine_local_rigid_correspondences_batched(corr, off)
    mask_valid_local = idx_all[keep]
    out = dict(mask_valid_local=mask_valid_local, idx_valid_src_refine=ivs[mask_valid_local],
               idx_valid_tgt_refine=None if idx_valid_tgt_refine is None else torch.as_tensor(idx_valid_tgt_refine, device=dev)[mask_valid_local],
               corres_3d_refine=corres_3d_refine[mask_valid_local, :],
               corres_3d_magnitude_refine=None if corres_3d_magnitude_refine is None else corres_3d_magnitude_refine[mask_valid_local],
               mask_robust=mask_robust, T_init=T_kabsch.to(torch.float32), patch_off=off)
    if icp_refine:
        # 3. one launch: Kabsch of the rows -> float32 values -> ICP(sources of the rows, targets of the rows) -> [s, T s]
        src, tgt = corr[:, :3].contiguous(), corr[:, 3:6].contiguous()
        r = engine.patch_loop(src, off, tgt, off, src, tgt, off, None, 0.0, 1e-6, max_corr_dist=float(icp_thres), max_iter=30,
                              rel_fitness=1e-6, rel_rmse=1e-6, icp_type="point2point", fixed_iters=False, min_corr=1,
                              init_round_f32=True, search=search)
        # the reference applies the FLOAT32 values of the ICP transform to the float32 sources (:1026-1030)
        T32 = r["T"].to(torch.float32)
        pid = torch.repeat_interleave(torch.arange(P, device=dev), rows_per_patch)
        moved = torch.einsum("nij,nj->ni", T32[pid, :3, :3], src) + T32[pid, :3, 3]
        rows = torch.cat([src, moved], dim=1)
        out.update(corres_3d_refine_apply_icp=rows,
                   corres_3d_magnitude_refine_apply_icp=torch.linalg.norm(rows[:, 3:6] - rows[:, :3], dim=1)[:, None],
                   T_icp=r["T"], fitness=r["fitness"], inlier_rmse=r["rmse"], iters=r["iters"])
    return out
