"""Mirror of the reference's src/piecewise_icp.py: `Piecewise_ICP(cfg)` with the same cfg keys and the same three
output files (src/piecewise_icp.py:203-216), plus the engine the north-star actually names.

cfg.engine (new key, default 'reference_octree'):

  'reference_octree'  what the reference function really does (src/piecewise_icp.py:115-199): octree
                      leaf centroids of both epochs, 1-NN matching of centroids, mean+std stability threshold, stable
                      cells keep their points, unstable cells are shifted by the centroid difference.  No ICP
                      iteration; `threshold` is read and not used, exactly like the reference (:41).
  'patch_icp'         the same octree leaves become patches; every patch gets a weighted-Kabsch initialisation from
                      1-NN pairs and a real point-to-point ICP (cfg.threshold = max correspondence distance) on the
                      MI355X, and its points are moved by the patch transform.

Open3D is not a dependency: the octree (o3d.geometry.Octree.convert_from_point_cloud with size_expand = 0 and its
depth-first `traverse` order) is restated with tensor ops on the GPU -- [3P-knowledge, parity unpinned]:
  cube = [origin, origin + size), size = largest bbox extent, centred on the bbox centre; points on the cube's max
  faces are not inserted; a leaf is a cell of edge size / 2^depth; children are visited in the order
  x + 2 y + 4 z, i.e. leaves come in Morton order; `f_traverse` skips every subtree whose internal node holds < 250
  points (:52) and keeps leaves with >= number_points_min points (:55-71).
"""
import copy
import os
import os.path as osp

import numpy as np

from .. import engine
from ..utils.ply import read_ply


def _dir_exist(path):
    os.makedirs(path, exist_ok=True)


def _octree_leaves(pts, origin, size, depth, n_min, internal_min=250):
    """pts (n,3) float64 cuda tensor.  Returns the kept leaves in depth-first order:
    (order (m,) point ids grouped by leaf, off (L+1,), centroids (L,3)); points outside the cube are dropped."""
    import torch
    dev = pts.device
    n = pts.shape[0]
    inb = ((pts >= origin) & (pts < origin + size)).all(dim=1)  # IsPointInBound
    ids = torch.nonzero(inb).squeeze(1)
    p = pts[ids]
    node_origin = origin.expand(p.shape[0], 3).clone()
    code = torch.zeros(p.shape[0], dtype=torch.int64, device=dev)
    child = float(size)
    for _ in range(depth):
        child *= 0.5
        bit = (p >= node_origin + child).to(torch.int64)  # x_index = point(0) < origin(0) + child_size ? 0 : 1
        code = code * 8 + bit[:, 0] + 2 * bit[:, 1] + 4 * bit[:, 2]
        node_origin = node_origin + bit.to(p.dtype) * child
    # every internal ancestor (root included) must hold >= internal_min points
    keep = torch.ones(p.shape[0], dtype=torch.bool, device=dev)
    for level in range(depth):  # level 0 = root ... depth-1 = parents of the leaves
        prefix = code // (8 ** (depth - level))
        _, inv, cnt = torch.unique(prefix, return_inverse=True, return_counts=True)
        keep &= cnt[inv] >= internal_min
    order = torch.argsort(code, stable=True)  # Morton order = depth-first child order; ids ascending inside a leaf
    code_s, ids_s, keep_s = code[order], ids[order], keep[order]
    uniq, inv, cnt = torch.unique_consecutive(code_s, return_inverse=True, return_counts=True)
    leaf_ok = cnt >= n_min
    first = torch.cumsum(cnt, 0) - cnt
    leaf_ok &= keep_s[first]
    sel = leaf_ok[inv]
    ids_k, inv_k = ids_s[sel], inv[sel]
    _, inv_c, cnt_c = torch.unique_consecutive(inv_k, return_inverse=True, return_counts=True)
    L = cnt_c.shape[0]
    off = torch.zeros(L + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(cnt_c, 0)
    cen = torch.zeros((L, 3), dtype=pts.dtype, device=dev)
    cen.index_add_(0, inv_c, pts[ids_k])
    cen /= cnt_c[:, None].to(pts.dtype)
    return ids_k, off, cen, n


def _nearest(query, ref):
    """1-NN of every query row among ref rows (both (m,3) float64 cuda) -> (idx, dist)."""
    import torch
    out_i, out_d = [], []
    for a in range(0, query.shape[0], 4096):
        d = torch.cdist(query[a:a + 4096], ref)
        dd, ii = d.min(dim=1)
        out_i.append(ii)
        out_d.append(dd)
    return torch.cat(out_i), torch.cat(out_d)


def Piecewise_ICP(cfg):
    """src/piecewise_icp.py:17-235.  Reads cfg.src_tile_overlap_path / tgt_tile_overlap_path (PLY), cfg.smax,
    cfg.number_points_min, cfg.threshold, cfg.output_root, cfg.tile_id, cfg.dataset, cfg.logging; writes
    results/piecewise_icp_dvfms_of_tile_{id}.txt (N x 4), piecewise_icp_dvfs_of_tile_{id}.txt (N x 6) and
    piecewise_dvfms_visualize_of_tile_{id}.txt."""
    import torch
    _lib_gpu = engine.require_gpu()  # no CPU fallback
    dev = torch.device("cuda")
    log = cfg.logging
    smax, n_min, threshold = cfg.smax, cfg.number_points_min, cfg.threshold
    mode = getattr(cfg, "engine", "reference_octree") if not isinstance(cfg, dict) else cfg.get("engine", "reference_octree")
    results = osp.join(cfg.output_root, 'results')
    _dir_exist(results)
    log.info('Start processing the current tile')

    src_np, _ = read_ply(cfg.src_tile_overlap_path)
    tgt_np, _ = read_ply(cfg.tgt_tile_overlap_path)
    src, tgt = torch.from_numpy(src_np).to(dev), torch.from_numpy(tgt_np).to(dev)

    # union bounding box, its 8 corners appended to both clouds so that both octrees share the same cells (:90-105)
    lo = torch.minimum(src.min(0).values, tgt.min(0).values)
    hi = torch.maximum(src.max(0).values, tgt.max(0).values)
    corners = torch.stack([torch.stack([(hi if (i >> a) & 1 else lo)[a] for a in range(3)]) for i in range(8)])
    src_all, tgt_all = torch.cat([src, corners]), torch.cat([tgt, corners])
    max_extent = float((hi - lo).max())
    depth = int(np.ceil(np.log2(max_extent / smax)))
    log.info("Octree depth: " + str(depth))
    centre = (lo + hi) / 2
    half = float((centre - lo).max())
    origin = torch.minimum(lo, centre - half)
    size = half * 2.0

    ids_s, off_s, cen_s, _ = _octree_leaves(src_all, origin, size, depth, n_min)
    log.info("Centroids found in source point cloud")
    ids_t, off_t, cen_t, _ = _octree_leaves(tgt_all, origin, size, depth, n_min)
    log.info("Centroids found in target point cloud")
    if cen_s.shape[0] == 0 or cen_t.shape[0] == 0:
        raise RuntimeError("no octree cell holds enough points (number_points_min / 250-point rule)")

    nn_idx, dist = _nearest(cen_s, cen_t)  # :142-148
    log.info("Corresponding centroids found")
    thr = dist.mean() + dist.std(unbiased=False)  # np.std, :154-156
    stable = dist <= thr
    log.info("Centroid pairs are categorized as stable. (" + str(int(stable.sum())) + "/" + str(dist.numel()) + ", " +
             str(np.round(float(stable.sum()) / dist.numel() * 100, 2)) + "%)")

    cnt_s = off_s[1:] - off_s[:-1]
    leaf_of_pt = torch.repeat_interleave(torch.arange(cen_s.shape[0], device=dev), cnt_s)
    pts_s = src_all[ids_s]

    if mode == 'reference_octree':
        # stable cells in np.unique(axis=0) order of their centroids (:166-171), then unstable cells in traversal order
        st = torch.nonzero(stable).squeeze(1)
        c = cen_s[st].cpu().numpy()
        lex = np.lexsort((c[:, 2], c[:, 1], c[:, 0])) if len(c) else np.zeros(0, np.int64)
        st = st[torch.from_numpy(lex).to(dev)]
        un = torch.nonzero(~stable).squeeze(1)
        dev_vec = cen_t[nn_idx] - cen_s  # :182-184
        dev_vec[stable] = 0.0  # stable cells keep their points (:174)
        # the points of the leaves in that order, one gather (the reference loops over the cells in Python, :170-193)
        leaves = torch.cat([st, un])
        cnt_l = cnt_s[leaves]
        new_off = torch.zeros(leaves.shape[0] + 1, dtype=torch.int64, device=dev)
        new_off[1:] = torch.cumsum(cnt_l, 0)
        total = int(new_off[-1])
        take = torch.repeat_interleave(off_s[:-1][leaves] - new_off[:-1], cnt_l, output_size=total) + torch.arange(total, device=dev)
        p = pts_s[take]
        dvfs = torch.cat([p, p + torch.repeat_interleave(dev_vec[leaves], cnt_l, dim=0, output_size=total)], dim=1).cpu().numpy()
        n_stable_pts = int(cnt_s[stable].sum())
    elif mode == 'patch_icp':
        # every kept source leaf is a patch; its target patch is the matched target leaf
        P = cen_s.shape[0]
        cnt_t = (off_t[1:] - off_t[:-1])[nn_idx]
        t_off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        t_off[1:] = torch.cumsum(cnt_t, 0)
        starts = off_t[:-1][nn_idx]
        gather = torch.repeat_interleave(starts - t_off[:-1], cnt_t) + torch.arange(int(t_off[-1]), device=dev)
        pts_t = tgt_all[ids_t][gather]
        shift = lo.clone()  # float32 kernels: work relative to the tile's lower corner
        s32, t32 = (pts_s - shift).float().contiguous(), (pts_t - shift).float().contiguous()
        eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
        T0 = eye.clone()
        T0[:, :3, 3] = cen_t[nn_idx] - cen_s  # centroid shift as the initial guess
        out = engine.piecewise_icp(s32, off_s, t32, t_off, init_T=T0, max_corr_dist=float(threshold), max_iter=30)
        rows6 = engine.apply_transform(s32, off_s, out["T"]).double()
        rows6 += torch.cat([shift, shift])
        dvfs = rows6.cpu().numpy()
        n_stable_pts = int(cnt_s[stable].sum())
    else:
        raise ValueError(f"unknown engine '{mode}'")

    mag = np.linalg.norm(dvfs[:, :3] - dvfs[:, 3:6], axis=1)
    dvfms = np.hstack((dvfs[:, :3], mag[:, None]))
    np.savetxt(osp.join(results, f'piecewise_icp_dvfms_of_tile_{cfg.tile_id}.txt'), dvfms)
    np.savetxt(osp.join(results, f'piecewise_icp_dvfs_of_tile_{cfg.tile_id}.txt'), dvfs)
    vis = copy.deepcopy(dvfms)
    if len(vis) > 1:
        vis[0, 3] = 0
        vis[1, 3] = {'rockfall': 0.06, 'brienz_tls': 5, 'mattertal': 10}.get(cfg.dataset, 10)  # :219-226
    np.savetxt(osp.join(results, f'piecewise_dvfms_visualize_of_tile_{cfg.tile_id}.txt'), vis)
    n_src = src_all.shape[0]
    log.info("Points in the source point cloud are categorized as stable. (" + str(n_stable_pts) + "/" + str(n_src) +
             ", " + str(np.round(n_stable_pts / n_src * 100, 2)) + "%)")
    return None
