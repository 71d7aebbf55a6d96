"""Mirror of the part of the reference's `src/functions.py` that sits next to the hot path.

    kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7, best_k=0, w_threshold=0)
        src/functions.py:12-85 (F2S3's weighted Kabsch; callers src/f2s3.py:340-366 and
        src/models/outlier_classifier.py:65-106).  One launch of f4l_kabsch2_batched for the whole batch.
    transformation_residuals(x1, x2, R, t)           src/functions.py:88-105
    point_cloud_tiling(config)                       src/functions.py:146-178 (the tiling front end, xy_tiling)
    robust_rigid_fit(corr, off, weights, coeff=1.0)  the arithmetic of `filter_input` AFTER the outlier network
        (src/models/outlier_classifier.py:65-106, called per supervoxel at src/f2s3.py:340-366), for all
        supervoxels of a tile at once; the network that produces `weights` is out of scope.

torch tensors on the GPU in, torch tensors out (dtype of x1).  Not differentiable (the reference's version is; its
callers on this path run under no_grad).  `best_k > 0` raises NotImplementedError: the reference applies batch element
0's selection to every element of the batch (:41-45), which is not reproduced.
"""
import os.path as osp

from .. import engine
from .._lib import require_gpu
from ..cpp_core.pcd_tiling.build import pcd_tiling


def transformation_residuals(x1, x2, R, t):
    """|| R x1_i + t - x2_i || per point: (b, n).  x1, x2 (b, n, 3); R (b, 3, 3); t (b, 3, 1)."""
    torch = require_gpu()
    moved = torch.matmul(R, x1.transpose(1, 2)) + t
    return torch.norm(moved - x2.transpose(1, 2), dim=1)


def kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7, best_k=0, w_threshold=0):
    torch = require_gpu()
    if best_k > 0:
        raise NotImplementedError("best_k is not supported (the reference reuses batch element 0's indices, src/functions.py:41-45)")
    if x1.dim() != 3 or x1.shape != x2.shape or x1.shape[2] != 3:
        raise ValueError("x1 and x2 must both be (b, n, 3)")
    b, n = x1.shape[0], x1.shape[1]
    off = torch.arange(b + 1, dtype=torch.int64, device=x1.device) * n
    w = None if weights is None else weights.reshape(b * n)
    R, t = engine.kabsch2_batched(x1.reshape(b * n, 3), x2.reshape(b * n, 3), off, w, normalize_w, float(w_threshold), eps)
    R, t = R.to(x1.dtype), t.to(x1.dtype).unsqueeze(2)
    res = transformation_residuals(x1, x2, R, t)
    return R, t, res, False


def point_cloud_tiling(config):
    """Tile the two raw epochs of `config` (src/functions.py:146-178): `xy_tiling` calls tile_point_clouds with the
    config's tile sizes and voxel size, projection axis chosen from the overlap box; `hv_tiling` and
    `python_based_tiling` are no-ops there too; anything else raises NotImplementedError."""
    src_pts_path = osp.join(config.data_dir, 'raw_pcd', config.src_name)
    tgt_pts_path = osp.join(config.data_dir, 'raw_pcd', config.tgt_name)
    if config.tiling_type == 'xy_tiling':
        pcd_tiling.tile_point_clouds(src_pts_path, tgt_pts_path, config.max_pts_per_tile, config.min_pts_per_tile,
                                     bool(config.voxel_size), config.voxel_size, 0.0, -1, config.tile_dir, config.verbose)
    elif config.tiling_type in ('hv_tiling', 'python_based_tiling'):
        return None
    else:
        raise NotImplementedError
    return None


def robust_rigid_fit(corr, off, weights, coeff=1.0, min_inliers=5, max_median=0.5, eps=1e-7):
    """Everything `OutlierClassifier.filter_input` does once it has the per-correspondence scores
    (src/models/outlier_classifier.py:70-103), batched over the ragged correspondence sets of a tile:

      1. weighted Kabsch #2 per set (`kabsch_transformation_estimation(x1, x2, weights)`, :72-73) and its residuals;
      2. inliers = residuals < coeff * median(residuals) (:79; torch.median = the LOWER median; coeff 2.5 for the
         Rockfall simulator, else 1, :75-78);
      3. where a set has >= 5 inliers and its median residual is < 0.5 (:90), the fit is repeated with weight 1 on the
         inliers and 0 elsewhere (:92-95) and the set is flagged `robust_estimate`.

    corr (n, 6) [x1 | x2] and weights (n,) on the GPU, off (P + 1,) int64.  Returns dict(rot_est (P, 3, 3), trans_est
    (P, 3, 1), robust_estimate (P,) bool, scores = weights, inliers (n,) bool, residuals (n,) of the final fit), in
    corr's dtype."""
    torch = require_gpu()
    if corr.dim() != 2 or corr.shape[1] != 6:
        raise ValueError("corr must be (n, 6)")
    n, P = corr.shape[0], off.shape[0] - 1
    x1, x2 = corr[:, :3].contiguous(), corr[:, 3:].contiguous()
    off = off.to(torch.int64)
    counts = off[1:] - off[:-1]
    pid = torch.repeat_interleave(torch.arange(P, device=corr.device), counts)

    def fit(w):
        R, t = engine.kabsch2_batched(x1, x2, off, w, True, 0.0, eps)
        R, t = R.to(corr.dtype), t.to(corr.dtype)
        res = torch.linalg.norm(torch.einsum("nij,nj->ni", R[pid], x1) + t[pid] - x2, dim=1)
        return R, t, res

    R, t, res = fit(weights.reshape(n).to(corr.dtype))
    # lower median of every set: sort by (set, residual), take element (len - 1) // 2 of the set's run
    order = torch.argsort(res, stable=True)
    order = order[torch.argsort(pid[order], stable=True)]
    pos = off[:-1] + torch.clamp(counts - 1, min=0) // 2
    med = torch.where(counts > 0, res[order[torch.clamp(pos, max=max(n - 1, 0))]], torch.full_like(res[:1], float("inf")).expand(P))
    inl = res < coeff * med[pid]
    n_inl = torch.zeros(P, dtype=torch.int64, device=corr.device).index_add_(0, pid, inl.to(torch.int64))
    robust = (n_inl >= min_inliers) & (med < max_median)
    if bool(robust.any()):
        # sets that are not re-fitted keep their first fit: give them their original weights in the second launch
        w2 = torch.where(robust[pid], inl.to(corr.dtype), weights.reshape(n).to(corr.dtype))
        R2, t2, res2 = fit(w2)
        R = torch.where(robust[:, None, None], R2, R)
        t = torch.where(robust[:, None], t2, t)
        res = torch.where(robust[pid], res2, res)
    return dict(rot_est=R, trans_est=t.unsqueeze(2), robust_estimate=robust, scores=weights, inliers=inl, residuals=res)
