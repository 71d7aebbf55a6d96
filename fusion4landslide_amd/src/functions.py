"""Mirror of the part of the reference's `src/functions.py` that sits next to the hot path.

    kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7, best_k=0, w_threshold=0)
        src/functions.py:12-85 (F2S3's weighted Kabsch; callers src/f2s3.py:340-366 and
        src/models/outlier_classifier.py:65-106).  One launch of f4l_kabsch2_batched for the whole batch.
    transformation_residuals(x1, x2, R, t)           src/functions.py:88-105
    point_cloud_tiling(config)                       src/functions.py:146-178 (the tiling front end, xy_tiling)

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
