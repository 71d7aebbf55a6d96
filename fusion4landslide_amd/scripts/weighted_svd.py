"""Mirror of the reference's scripts/weighted_svd.py (same names, signatures and return conventions), computed
by the batched HIP Kabsch kernel (f4l_kabsch_batched) instead of ~12 torch ops + torch.svd per call.

    weighted_procrustes(src_points, ref_points, weights=None, weight_thresh=0.0, eps=1e-7, return_transform=True, ...)
    weighted_svd(src_pts, tgt_pts, eps=1e-6, weights=None, weight_thresh=0.0, return_transform=True)
    refine_local_rigid_correspondences(corr_neigh_2, refine_type='SVD', weights=None)
    refine_local_rigid_correspondences_batched(corr, off, weights=None)     (new: all patches of a tile at once)
"""
from .. import engine


def _batched(src, ref, weights, weight_thresh, eps):
    import torch
    squeeze = src.ndim == 2
    if squeeze:
        src, ref = src.unsqueeze(0), ref.unsqueeze(0)
        if weights is not None:
            weights = weights.unsqueeze(0)
    B, N = src.shape[0], src.shape[1]
    dt = src.dtype if src.dtype in (torch.float32, torch.float64) else torch.float32
    off = torch.arange(0, (B + 1) * N, N, dtype=torch.int64, device=src.device)
    w = None if weights is None else weights.reshape(B * N).to(dt)
    R, t = engine.kabsch_batched(src.reshape(B * N, 3).to(dt), ref.reshape(B * N, 3).to(dt), off, w, weight_thresh, eps)
    return R.to(dt), t.to(dt), squeeze, B


def _finish(R, t, squeeze, B, return_transform):
    import torch
    if return_transform:
        T = torch.eye(4, dtype=R.dtype, device=R.device).unsqueeze(0).repeat(B, 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3] = t
        return T.squeeze(0) if squeeze else T
    if squeeze:
        return R.squeeze(0), t.squeeze(0)
    return R, t


def weighted_procrustes(src_points, ref_points, weights=None, weight_thresh=0.0, eps=1e-7, return_transform=True,
                        return_rmse=True):
    """scripts/weighted_svd.py:58-129.  (B,N,3) or (N,3) tensors on the GPU; weights (B,N) / (N,) or None.
    Returns (B,4,4)/(4,4) when return_transform else (R, t)."""
    R, t, squeeze, B = _batched(src_points, ref_points, weights, weight_thresh, eps)
    return _finish(R, t, squeeze, B, return_transform)


def weighted_svd(src_pts, tgt_pts, eps=1e-6, weights=None, weight_thresh=0.0, return_transform=True):
    """scripts/weighted_svd.py:10-55 (older variant, weights shaped (B,N,1); no caller in the reference)."""
    if weights is not None and weights.ndim == src_pts.ndim:
        weights = weights.squeeze(-1)
    R, t, squeeze, B = _batched(src_pts, tgt_pts, weights, weight_thresh, eps)
    return _finish(R, t, squeeze, B, return_transform)


def refine_local_rigid_correspondences(corr_neigh_2, refine_type='SVD', weights=None):
    """scripts/weighted_svd.py:132-159: Kabsch with eps=1e-6, drop rows whose residual is >= 1 m (:145-147);
    returns (pruned (n',6), transform (4,4) float32 on the GPU)."""
    import torch
    if refine_type != 'SVD':
        # the reference's 'RANSAC' branch leaves its result undefined (UnboundLocalError, :152-159)
        raise NotImplementedError("only refine_type='SVD' is implemented")
    rot, tra = weighted_procrustes(corr_neigh_2[:, :3], corr_neigh_2[:, 3:6], weights=weights, weight_thresh=0.0,
                                   eps=1e-6, return_transform=False)
    off = torch.tensor([0, corr_neigh_2.shape[0]], dtype=torch.int64, device=corr_neigh_2.device)
    res = engine.kabsch_residuals(corr_neigh_2[:, :3], corr_neigh_2[:, 3:6], off, rot.unsqueeze(0), tra.unsqueeze(0))
    pruned = corr_neigh_2[res < 1.0, :]
    T = torch.eye(4, device=corr_neigh_2.device)
    T[:3, :3] = rot
    T[:3, 3] = tra
    return pruned, T


def refine_local_rigid_correspondences_batched(corr, off, weights=None, max_res=1.0):
    """All patches of a tile in one launch: corr (n,6) rows grouped by `off` (P+1,).
    Returns (keep mask (n,) bool, T (P,4,4) float64)."""
    import torch
    src, ref = corr[:, :3].contiguous(), corr[:, 3:6].contiguous()
    R, t = engine.kabsch_batched(src, ref, off, weights, 0.0, 1e-6)
    res = engine.kabsch_residuals(src, ref, off, R, t)
    P = off.shape[0] - 1
    T = torch.eye(4, dtype=torch.float64, device=corr.device).unsqueeze(0).repeat(P, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return res < max_res, T
