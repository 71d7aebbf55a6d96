"""Mirror of the two helpers of the reference's src/rgb_guided.py that sit on the per-patch rigid-fit path (the rest of that
file is image lifting and matching: out of scope).

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
