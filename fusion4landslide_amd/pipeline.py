"""The full hot path of one tile, end to end on the device (BASELINE.json configs[4], "Full fusion hot path: HIP supervoxel seg
+ piecewise ICP + weighted-SVD"), as the reference's `implement_c2f_matching` strings it together for its 3D-only mode
(src/coarse_to_fine_matching.py:201-290) -- minus the learned feature matching, whose place (producing point
matches and patch matches) is taken here by nearest neighbours:

    _compute_median_resolution            base:2716-2754   engine.median_resolution (exact 2-NN, f4l_knn)
    implement_partition                   base:2658-2694   supervoxel partition of the source epoch, resolution =
                                                           max(sqrt(3) * 10 * median_res, voxel) (:2668-2671)
                                                           (f4l_supervoxel: the reference's labels, the default; or the
                                                            parallel variant f4l_supervoxel_parallel)
    load_partition / prepare_pts2spt_dict base:1237-1332   f4l_labels_to_csr + f4l_gather_points
    (patch matches: every target point joins the patch of its nearest source point -- f4l_nn_query; point matches: 1-NN of
     each source point inside its target patch within 2 x icp_threshold -- f4l_nn_refine at the identity)
    fine_matching_with_different_types    base:3236-3436   f4l_patch_loop (Kabsch -> ICP -> rows) + f4l_nn_refine

Every stage is timed with events on the launch stream; nothing leaves the device between stages except the two counts the
reference also materialises (the number of supervoxels and, inside f4l_knn, the grid size).
"""
import numpy as np

from . import engine


def full_path(src, tgt, k=30, icp_threshold=0.1, voxel_size=0.0, partition="identical", max_iter=30, fixed_iters=False,
              search="f64", resolution=None, keep_inputs=False):
    """src, tgt: (n, 3) float32 CUDA tensors (two epochs of one tile).  `partition`: "identical" (default, like the entry points:
    the reference's own supervoxel labels, f4l_supervoxel) or "parallel" (the all-device variant: same K and criteria, other
    labels).  Returns dict(rows (n_src, 6) dense displacement
    rows in patch order, sparse (m, 6), labels, K, T, fitness, rmse, iters, order (patch-contiguous source order),
    resolution, stage_ms {name: milliseconds}[, with keep_inputs: what the per-patch loop was given -- patch_src, patch_tgt (the
    two epochs in patch order), corr_src, corr_ref, corr_off (the point matches of every patch)])."""
    torch = engine.require_gpu()
    stages, marks = [], []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        stages.append(name)

    mark("start")
    if resolution is None and partition == "parallel" and k >= 2:
        # the partition's neighbour search does not depend on the resolution: run it first and let it serve the median point
        # spacing of the source epoch too (slot 1 of every row), instead of a 2-NN pass of its own over the same cloud
        # (f4l_partition_neighbours / f4l_partition_segment: the segmentation adopts the search's order; k beyond the lane-per-query
        #  search falls back to the two ordinary calls)
        nb = knn_idx = None
        if k <= 36:
            nb = engine.partition_neighbours(src, k, return_nn1=True)
            nn1 = nb.nn1_d2
        else:
            knn_idx, nrm, nn1 = engine.knn_normals(src, k, return_nn1=True)
        mark("neighbours")
        # the second epoch is binned once for both of its searches: nearest other target point (the median) and nearest source
        # point (the patch it joins once the labels exist) -- f4l_epoch_join
        tgt_nn, tgt_nn1 = engine.epoch_join(src, tgt)
        med = engine.median_resolution(src, tgt, src_nn1_d2=nn1, tgt_nn1_d2=tgt_nn1)
        resolution = max(np.sqrt(3.0) * 10.0 * med, float(voxel_size), 1e-6)  # base:2668-2671
        del tgt_nn1
        mark("median_resolution")
        if nb is not None:
            labels, info = engine.partition_segment(nb, float(resolution))
        else:
            labels, info = engine.supervoxel_segment_device(src, nrm, knn_idx, float(resolution))
        info_h = info.cpu()
        K = int(info_h[0])
        if int(info_h[2]) & 14:
            raise RuntimeError(f"f4l_partition_segment: the segmentation did not finish (status bits {int(info_h[2])})")
        del nb, knn_idx, nn1
        mark("supervoxel_partition")
    else:
        tgt_nn = None
        if resolution is None:
            med = engine.median_resolution(src, tgt)
            resolution = max(np.sqrt(3.0) * 10.0 * med, float(voxel_size), 1e-6)  # base:2668-2671
        else:
            med = float(resolution) / (np.sqrt(3.0) * 10.0)
        mark("median_resolution")
        labels, K = (engine.supervoxel_parallel if partition == "parallel" else engine.supervoxel)(src, k, float(resolution))
        mark("supervoxel_partition")
    st = _patches_and_registration(torch, src, tgt, labels, tgt_nn, K, med, icp_threshold, max_iter, fixed_iters, search, mark, keep_inputs)
    torch.cuda.synchronize()
    ms = {stages[i]: marks[i - 1].elapsed_time(marks[i]) for i in range(1, len(stages))}
    ms["total"] = marks[0].elapsed_time(marks[-1])
    return dict(labels=labels, K=K, resolution=float(resolution), stage_ms=ms, **st)


def _patches_and_registration(torch, src, tgt, labels, tgt_nn, K, med, icp_threshold, max_iter, fixed_iters, search, mark, keep_inputs=False):
    """The stages after the partition, on one device: patches of both epochs (a target point joins the patch of its nearest
    source point: `tgt_nn` (m,) int32 / int64 indices into `src` when the caller has them already, else f4l_nn_query), point matches,
    the per-patch loop, the refinement.  Returns dict(rows, sparse, T, fitness, rmse, iters, order, src_off, tgt_off)."""
    order_s, off_s = engine.labels_to_csr(labels, K)
    if tgt_nn is None:
        tgt_nn = engine.nn_query(src, tgt, 1)[:, 0]
    if tgt_nn.dtype == torch.int32:
        order_t, off_t = engine.labels_to_csr_via(labels, tgt_nn.contiguous(), K)
    else:
        order_t, off_t = engine.labels_to_csr(labels[tgt_nn], K)
    ps, pt = engine.gather_points(src, order_s), engine.gather_points(tgt, order_t)
    mark("patches")
    P = K
    dev = src.device
    eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
    max_t = int((off_t[1:] - off_t[:-1]).max().item()) if P else 0
    max_s = int((off_s[1:] - off_s[:-1]).max().item()) if P else 0
    m, _ = engine.nn_refine(ps, off_s, pt, off_t, eye, torch.full((P,), 2.0 * icp_threshold, dtype=torch.float64, device=dev),
                            max_tgt_patch=max_t, return_rows=False)
    # rows that found a match and their targets as the loop's correspondence lists: one running count, one pass (f4l_match_lists)
    cs, ct, coff = engine.match_lists(ps, off_s, pt, off_t, m)
    mark("point_matches")
    out = engine.patch_loop(ps, off_s, pt, off_t, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=icp_threshold,
                            max_iter=max_iter, fixed_iters=fixed_iters, max_src_patch=max_s, max_tgt_patch=max_t, search=search)
    mark("patch_loop")
    thr = torch.clamp(2.0 * out["rmse"], min=med)  # base:3420-3424
    thr = torch.where(torch.isfinite(thr), thr, torch.full_like(thr, med))
    nn2, sparse = engine.nn_refine(ps, off_s, pt, off_t, out["T"], thr, max_tgt_patch=max_t)
    sparse = sparse[nn2 >= 0]
    mark("nn_refine")
    res = dict(rows=out["rows"], sparse=sparse, T=out["T"], fitness=out["fitness"], rmse=out["rmse"], iters=out["iters"],
               order=order_s, src_off=off_s, tgt_off=off_t)
    if keep_inputs:
        res.update(patch_src=ps, patch_tgt=pt, corr_src=cs, corr_ref=ct, corr_off=coff)
    return res


def full_path_slabs(local_src, local_gid, local_tgt, dist, rank, world, halo, resolution, k=30, icp_threshold=0.1, max_iter=30,
                    fixed_iters=False, search="f64"):
    """The same path for ONE cloud spread over `world` GPUs (BASELINE.json configs[4]): the partition by slabs
    along x with a halo (slabs.slab_supervoxel), the second epoch joined to the slabs' patches (slabs.slab_targets), then
    patches, point matches, the per-patch loop and the refinement on the rank that owns the patches -- a supervoxel never
    crosses a cut, so nothing after the partition needs another exchange; the caller gathers what it wants of the per-patch
    results (sharding.PatchResultGather).  local_src / local_tgt: this rank's arbitrary chunks of the two epochs (float32
    CUDA tensors), local_gid (n,) int64 the source chunk's global point ids.  `resolution` is the supervoxel resolution
    (base:2668-2671; the median point spacing it derives from is a property of the whole cloud: compute it on a tile).
    Returns dict(rows (n_own, 6) in patch order, gid (n_own,) the global ids of those rows' source points, T, fitness, rmse,
    iters per LOCAL patch, K_local, K_total, offset (global patch id = offset + local), sparse, n_uncertified (neighbour
    lists + target joins of the whole job that the halo could not certify: widen the halo if not 0), stage_ms)."""
    torch = engine.require_gpu()
    from . import slabs
    stages, marks = [], []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        stages.append(name)

    mark("start")
    med = float(resolution) / (np.sqrt(3.0) * 10.0)
    sv = slabs.slab_supervoxel(local_src, local_gid, k, float(resolution), dist, rank, world, halo)
    mark("supervoxel_partition")
    tg = slabs.slab_targets(local_tgt, sv, dist, rank, world, halo)
    mark("target_exchange")
    K = sv["K_local"]
    if K == 0:
        raise ValueError("this rank's slab holds no source point: use fewer ranks")
    st = _patches_and_registration(torch, sv["xyz"].contiguous(), tg["xyz"], sv["labels_local"].to(torch.int32), tg["nn"], K, med, icp_threshold,
                                   max_iter, fixed_iters, search, mark)
    torch.cuda.synchronize()
    ms = {stages[i]: marks[i - 1].elapsed_time(marks[i]) for i in range(1, len(stages))}
    ms["total"] = marks[0].elapsed_time(marks[-1])
    return dict(gid=sv["gid"][st["order"].to(torch.int64)], K_local=K, K_total=sv["K_total"], offset=sv["offset"],
                n_uncertified=sv["n_uncertified"] + tg["n_uncertified"], n_halo=sv["n_halo"], n_forwarded=tg["n_forwarded"],
                resolution=float(resolution), stage_ms=ms, **st)
