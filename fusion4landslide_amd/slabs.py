"""Multi-GPU split of the PARTITION stage (kNN + normals + supervoxels) of one large cloud: slabs along x with a halo
(BASELINE.json configs[4]).  The per-patch loop shards by patches (sharding.py, no data-path collective); the
partition is the one stage of the path with a real exchange step, because a point's neighbours can lie on another GPU.

One process per GPU (`torch.distributed`: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).  Every rank
starts with an arbitrary chunk of the cloud (e.g. its share of the file) and ends with the supervoxel labels of the points it
OWNS.  Steps, with their collectives:

  1. grid     all_reduce(MIN / MAX) of the bounding box; the resolution grid of `GridSample`
              (codelibrary/geometry/point_cloud/grid_sample.h:48-68) is anchored at the WHOLE cloud's minimum.
  2. cuts     histogram of the points per x-column of that grid, all_reduce(SUM); the slabs are runs of whole columns with
              balanced point counts, so every grid cell lies in exactly one slab and the slabs' supervoxel targets (occupied
              cells) add up to the whole cloud's.
  3. owners   all_to_all (variable sizes) of the points to the rank that owns their column.
  4. halo     each rank sends its two neighbours the owned points within `halo` metres of the common cut (point-to-point
              volume only: on the xGMI mesh neighbours are directly linked) -- realised as the same variable all_to_all with
              empty messages for non-neighbours.
  5. kNN      exact kNN + PCA normals of owned + halo points on the rank's GPU (f4l_knn_normals).  An owned point's list is
              the WHOLE cloud's list when its k-th distance does not reach past the halo's outer edge; the count of points
              for which that fails is all_reduced and returned (0 for a halo of a few neighbour spacings; the caller widens
              the halo otherwise).
  6. segment  supervoxels of the owned points only (f4l_supervoxel_segment_device, neighbours in the halo dropped: a
              supervoxel never crosses a cut -- the seam is a straight supervoxel boundary along a grid line), counted in the
              whole cloud's grid.
  7. labels   all_gather of the per-slab counts; global label = exclusive prefix + local label: 0 .. K-1, K = the whole
              cloud's occupied cells.

The SECOND epoch joins the patches afterwards (`slab_targets`): every target point goes to the rank that owns its grid column,
which finds its nearest source point among the slab's owned + halo source points (exact when that distance does not reach
past the halo's outer edge; counted like in step 5); a target point whose nearest source point is a halo point is forwarded
to the neighbour that owns that point (point-to-point volume again).  Each target point ends on exactly one rank -- the one
that owns the patch it joins -- so the per-patch loop runs where the patch lives, with no re-sharding of patches and no
further collective until the per-patch results are gathered (pipeline.full_path_slabs).

If the neighbour graph proves fragile for a data set (e.g. slabs thinner than the halo), the fallback SURVEY.md 8(e) names
stands: tiles (<= 1 M points, cpp_core/pcd_tiling) are independent, one replica of the single-GPU partition per tile.
"""
import numpy as np


def _staged(dist, t):
    """gloo moves host tensors only: device tensors are staged through the host there (tests and single-GPU dry runs of the
    N > 1 path; on a GPU node the backend is "nccl" = RCCL and the tensors stay on the device)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(dist, t, op):
    # the NCCL (= RCCL) process group has no bitwise reductions ("Cannot use ReduceOp.BOR with NCCL"): refuse them wherever the
    # backend is not gloo, so that a path that only ever ran over gloo cannot carry one onto the GPUs
    if op in (dist.ReduceOp.BOR, dist.ReduceOp.BAND, dist.ReduceOp.BXOR) and dist.get_backend() != "gloo":
        raise ValueError(f"{op} is not available on the {dist.get_backend()} backend: reduce bits with MAX / SUM instead")
    if _staged(dist, t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def _all_to_all(dist, out, inp, out_splits=None, in_splits=None):
    if _staged(dist, inp):
        ho = out.cpu()
        dist.all_to_all_single(ho, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


def _a2a_v(dist, torch, payload, dest, world):
    """Variable-size all_to_all of the rows of `payload` (n, c) float64 to ranks `dest` (n,) int64.  Returns the received rows
    (ordered by source rank, original order inside a source)."""
    order = torch.argsort(dest, stable=True)
    send = payload[order].contiguous()
    counts = torch.bincount(dest, minlength=world).to(torch.int64)
    if world == 1:
        return send
    both = torch.zeros(2 * world, dtype=torch.int64, device=payload.device)  # [what I send to r | what r sends to me]
    both[:world] = counts
    _all_to_all(dist, both[world:], both[:world])
    # the split sizes of the payload exchange are host integers: ONE read-back of both halves (through a pinned buffer when the
    # tensors live on a GPU), not one `.tolist()` per half
    if both.is_cuda:
        host = torch.empty(2 * world, dtype=torch.int64, pin_memory=True)
        host.copy_(both, non_blocking=True)
        torch.cuda.current_stream(both.device).synchronize()
    else:
        host = both
    c = payload.shape[1]
    sizes = (host * c).tolist()
    recv = torch.empty((sum(sizes[world:]) // c, c), dtype=payload.dtype, device=payload.device)
    _all_to_all(dist, recv.view(-1), send.view(-1), sizes[world:], sizes[:world])
    return recv


_FLAG_BITS = 16


def _raise_together(dist, torch, dev, world, flag, what):
    """Every rank raises when ANY rank reports `flag` != 0: the OR of the ranks' flags (small non-negative bit masks) as ONE
    all_reduce(MAX) over their bits -- RCCL has no bitwise reduction (ReduceOp.BOR is gloo only)."""
    flag = int(flag)
    if flag < 0 or flag >= 1 << _FLAG_BITS:
        raise ValueError("flag must be a bit mask below 2^%d" % _FLAG_BITS)
    f = torch.tensor([(flag >> b) & 1 for b in range(_FLAG_BITS)], dtype=torch.int64, device=dev)
    if world > 1:
        _all_reduce(dist, f, dist.ReduceOp.MAX)
    allf = sum(int(v) << b for b, v in enumerate(f.tolist()))
    if allf:
        raise RuntimeError(f"{what} [flags over all ranks: {allf}]")


def plan_slabs(local_xyz, resolution, dist, world):
    """Steps 1-2.  local_xyz (n, 3) float32 torch tensor (any device).  Returns dict(grid_min, grid_max float32 (3,) numpy,
    bounds (world + 1,) int64 numpy: slab r owns the grid columns bounds[r] .. bounds[r + 1] - 1, n_total)."""
    import torch
    dev = local_xyz.device
    big = torch.finfo(torch.float32).max
    mn = local_xyz.min(dim=0).values if local_xyz.shape[0] else torch.full((3,), big, device=dev)
    mx = local_xyz.max(dim=0).values if local_xyz.shape[0] else torch.full((3,), -big, device=dev)
    if world > 1:
        _all_reduce(dist, mn, dist.ReduceOp.MIN)
        _all_reduce(dist, mx, dist.ReduceOp.MAX)
    gmin, gmax = mn.double(), mx.double()
    ncol = int((gmax[0] - gmin[0]) / resolution + 1)  # grid_sample.h:49
    col = torch.clamp(((local_xyz[:, 0].double() - gmin[0]) / resolution).to(torch.int64), 0, ncol - 1)
    hist = torch.bincount(col, minlength=ncol).to(torch.int64)
    if world > 1:
        _all_reduce(dist, hist, dist.ReduceOp.SUM)
    cum = torch.cumsum(hist, 0).cpu().numpy()
    n_total = int(cum[-1])
    bounds = np.zeros(world + 1, dtype=np.int64)
    for r in range(1, world):
        bounds[r] = max(bounds[r - 1], int(np.searchsorted(cum, r * n_total / world, side="left")) + 1)
    bounds[world] = ncol
    bounds = np.minimum(bounds, ncol)
    return dict(grid_min=mn.cpu().numpy(), grid_max=mx.cpu().numpy(), bounds=bounds, n_total=n_total, ncol=ncol, resolution=float(resolution))


def slab_supervoxel(local_xyz, local_gid, k, resolution, dist, rank, world, halo, knn_normals_fn=None, segment_fn=None):
    """Steps 3-7 (and 1-2 through plan_slabs).  local_xyz (n, 3) float32, local_gid (n,) int64 global point ids of this rank's
    chunk.  knn_normals_fn(xyz) -> (idx (m, k) int, d2 (m, k) float64, normals (m, 3) float64) and segment_fn(xyz, normals, knn,
    resolution, grid_bbox) -> (labels (m,) int, K) default to the HIP path (f4l_knn_normals, f4l_supervoxel_segment_device);
    the CPU tests inject checkers.  Returns dict(xyz, gid, labels (global), knn_gid (n_owned, k) neighbour GLOBAL ids, d2,
    normals, K_local, K_total, offset, n_uncertified (whole job), plan)."""
    import torch
    dev = local_xyz.device
    plan = plan_slabs(local_xyz, resolution, dist, world)
    b = plan["bounds"]
    x0 = float(plan["grid_min"][0])
    cuts = x0 + b.astype(np.float64) * resolution  # slab r = [cuts[r], cuts[r + 1]) in x (the last one closed by the clamp)
    width = np.diff(cuts)
    # (decided from `bounds`, which every rank holds identically: all ranks raise together, nobody is left in a collective)
    if world > 1 and (width <= 0).any():
        empty = [int(r) for r in np.nonzero(width <= 0)[0]]
        raise ValueError(f"slab(s) {empty} own no grid column (one column of the resolution grid holds more than 1/{world} of the points; "
                         f"column bounds {b.tolist()}): halos only travel between adjacent ranks, so an empty slab would cut its "
                         "neighbours off from each other -- use fewer ranks or tiles")
    if world > 1 and halo > width.min():
        raise ValueError(f"halo {halo} m is wider than the thinnest slab ({width.min():.3f} m): use fewer ranks or tiles")
    col = torch.clamp(((local_xyz[:, 0].double() - x0) / resolution).to(torch.int64), 0, plan["ncol"] - 1)
    owner = torch.bucketize(col, torch.from_numpy(b[1:-1]).to(dev), right=True) if world > 1 else torch.zeros_like(col)
    packed = torch.cat([local_xyz.double(), local_gid.double()[:, None]], dim=1)  # (ids < 2^53 travel exactly as doubles)
    own = _a2a_v(dist, torch, packed, owner, world)
    # halo: owned points within `halo` of a cut go to the rank on the other side
    x = own[:, 0]
    parts, dests = [], []
    if rank > 0:
        m = x < cuts[rank] + halo
        parts.append(own[m]); dests.append(torch.full((int(m.sum()),), rank - 1, dtype=torch.int64, device=dev))
    if rank < world - 1:
        m = x >= cuts[rank + 1] - halo
        parts.append(own[m]); dests.append(torch.full((int(m.sum()),), rank + 1, dtype=torch.int64, device=dev))
    out = torch.cat(parts) if parts else own[:0]
    dst = torch.cat(dests) if dests else torch.zeros(0, dtype=torch.int64, device=dev)
    halo_pts = _a2a_v(dist, torch, out, dst, world) if world > 1 else own[:0]
    n_own = own.shape[0]
    allp = torch.cat([own, halo_pts])
    xyz_all = allp[:, :3].float().contiguous()
    gid_all = allp[:, 3].to(torch.int64)

    if knn_normals_fn is None:
        from . import engine

        def knn_normals_fn(p):
            idx, nrm, d2 = engine.knn_normals(p, k, return_d2=True)
            return idx, d2, nrm
    # failures are decided TOGETHER: a rank that raised alone would leave the others waiting in the next collective
    _raise_together(dist, torch, dev, world, 0 if xyz_all.shape[0] > k else 1,
                    f"a slab holds {xyz_all.shape[0]} points (owned + halo), not more than k = {k}: use fewer ranks")
    idx, d2, nrm = knn_normals_fn(xyz_all)
    idx, d2, nrm = idx[:n_own].to(torch.int64), d2[:n_own], nrm[:n_own]
    # certified: the k-th neighbour lies inside slab + halo whatever lies beyond
    dk = torch.sqrt(d2[:, -1])
    xo = own[:, 0]
    ok = torch.ones(n_own, dtype=torch.bool, device=dev)
    if rank > 0:
        ok &= dk < xo - (cuts[rank] - halo)
    if rank < world - 1:
        ok &= dk < (cuts[rank + 1] + halo) - xo
    bad = torch.tensor([int((~ok).sum())], dtype=torch.int64, device=dev)
    if world > 1:
        _all_reduce(dist, bad, dist.ReduceOp.SUM)

    # segmentation of the owned points; neighbours in the halo are "no neighbour" there
    knn_local = torch.where(idx < n_own, idx, torch.full_like(idx, -1))
    box = np.concatenate([plan["grid_min"], plan["grid_max"]]).astype(np.float32)
    if segment_fn is None:
        from . import engine

        def segment_fn(p, normals, knn_, res, grid_bbox):
            labels, info = engine.supervoxel_segment_device(p, normals, knn_.to(torch.int32), res, grid_bbox=grid_bbox)
            info = info.cpu()
            return labels, int(info[0]), int(info[2])
    status = 0
    if n_own:
        seg = segment_fn(xyz_all[:n_own].contiguous(), nrm, knn_local, resolution, box)
        labels, K_local = seg[0], seg[1]
        status = int(seg[2]) if len(seg) > 2 else 0
    else:
        labels, K_local = torch.zeros(0, dtype=torch.int64, device=dev), 0
    # status bits of f4l_supervoxel_segment_device other than 1 (a disconnected neighbour graph stopped above its target: a
    # valid partition) mean the labels are not a finished segmentation: lambda schedule exhausted (2), sweep budget hit (4)
    _raise_together(dist, torch, dev, world, status & ~1,
                    "the device segmentation of a slab did not finish (status bits: 2 lambda schedule exhausted, 4 sweep budget hit)")
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = K_local
    if world > 1:
        _all_reduce(dist, counts, dist.ReduceOp.SUM)
    offset = int(counts[:rank].sum())
    if world > 1 and bool((counts == 0).any()):  # (from the reduced counts: every rank raises, nobody is left in a later collective)
        raise ValueError(f"slab(s) {[int(r) for r in torch.nonzero(counts == 0).flatten().tolist()]} hold no source point: use fewer ranks")
    return dict(xyz=xyz_all[:n_own], gid=gid_all[:n_own], labels=labels.to(torch.int64) + offset, knn_gid=gid_all[idx], d2=d2, normals=nrm,
                K_local=int(K_local), K_total=int(counts.sum()), offset=offset, n_uncertified=int(bad.item()), plan=plan, n_halo=int(halo_pts.shape[0]),
                xyz_all=xyz_all, gid_all=gid_all, n_own=n_own, cuts=cuts, labels_local=labels.to(torch.int64))


def slab_targets(local_tgt, sv, dist, rank, world, halo, nn_fn=None):
    """The second epoch on the slabs of `sv` (the dict slab_supervoxel returned for the first).  local_tgt (m, 3) float32: this
    rank's arbitrary chunk of the target epoch.  nn_fn(cloud, queries) -> (idx (q,) int, d2 (q,) float64) defaults to the HIP
    path (f4l_nn_query, k = 1).  Returns dict(xyz (t, 3) float32: the target points whose nearest source point this rank OWNS,
    nn (t,) int64: that point's index among the owned source points (sv["xyz"]), d2 (t,) float64, n_forwarded (sent to a
    neighbour), n_uncertified (whole job))."""
    import torch
    dev = local_tgt.device
    plan, cuts, n_own = sv["plan"], sv["cuts"], sv["n_own"]
    b = plan["bounds"]
    x0 = float(plan["grid_min"][0])
    col = torch.clamp(((local_tgt[:, 0].double() - x0) / plan["resolution"]).to(torch.int64), 0, plan["ncol"] - 1)
    owner = torch.bucketize(col, torch.from_numpy(b[1:-1]).to(dev), right=True) if world > 1 else torch.zeros_like(col)
    own_t = _a2a_v(dist, torch, local_tgt.double(), owner, world).float().contiguous()
    if nn_fn is None:
        from . import engine

        def nn_fn(cloud, queries):
            idx, d2 = engine.nn_query(cloud, queries, 1, return_d2=True)
            return idx[:, 0].to(torch.int64), d2[:, 0]
    if own_t.shape[0] and sv["xyz_all"].shape[0]:
        idx, d2 = nn_fn(sv["xyz_all"], own_t)
        idx = idx.to(torch.int64)
    else:  # (a slab without source points keeps no target point: there is no patch here to join)
        idx = torch.full((own_t.shape[0],), -1, dtype=torch.int64, device=dev)
        d2 = torch.full((own_t.shape[0],), float("inf"), dtype=torch.float64, device=dev)
    # exact while the nearest source point lies inside slab + halo whatever lies beyond
    d = torch.sqrt(d2)
    x = own_t[:, 0].double()
    ok = idx >= 0
    if rank > 0:
        ok &= d < x - (cuts[rank] - halo)
    if rank < world - 1:
        ok &= d < (cuts[rank + 1] + halo) - x
    bad = torch.tensor([int((~ok).sum())], dtype=torch.int64, device=dev)
    if world > 1:
        _all_reduce(dist, bad, dist.ReduceOp.SUM)
    mine = (idx >= 0) & (idx < n_own)
    keep_xyz, keep_nn, keep_d2 = own_t[mine], idx[mine], d2[mine]
    n_fwd = 0
    if world > 1:
        # the nearest source point is a halo point: the target point joins a patch of the neighbour that owns it
        away = idx >= n_own
        a_idx = idx[away]
        a_x = sv["xyz_all"][a_idx, 0].double()
        dest = torch.where(a_x < cuts[rank], torch.full_like(a_idx, rank - 1), torch.full_like(a_idx, rank + 1))
        payload = torch.cat([own_t[away].double(), sv["gid_all"][a_idx].double()[:, None], d2[away][:, None]], dim=1)
        n_fwd = int(away.sum())
        got = _a2a_v(dist, torch, payload, dest, world)
        if got.shape[0]:
            order = torch.argsort(sv["gid"])
            pos = torch.searchsorted(sv["gid"][order], got[:, 3].to(torch.int64))
            local = order[torch.clamp(pos, max=max(n_own - 1, 0))]
            assert bool((sv["gid"][local] == got[:, 3].to(torch.int64)).all()), "forwarded target points must name a source point owned here"
            keep_xyz = torch.cat([keep_xyz, got[:, :3].float()])
            keep_nn = torch.cat([keep_nn, local])
            keep_d2 = torch.cat([keep_d2, got[:, 4]])
    return dict(xyz=keep_xyz.contiguous(), nn=keep_nn, d2=keep_d2, n_forwarded=n_fwd, n_uncertified=int(bad.item()))
