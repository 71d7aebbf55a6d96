"""Mirror of the hot-path part of the reference's utils/o3d_tools.py, same names and argument meaning:

    icp_registration(src_pcd, tgt_pcd, initial_transform, threshold=0.1, icp_type='point2point') -> dict
    get_correspondence_pairwise_point_clouds, array2pcd, tensor2pcd, pcd2array, array2tensor, pcd2tensor

`icp_registration` (utils/o3d_tools.py:12-71) is a single-patch call into the batched HIP kernel
(f4l_piecewise_icp); use `fusion4landslide_amd.engine.piecewise_icp` to run all patches of a tile in one
launch.  Visualisation helpers (utils/o3d_tools.py:259-507), colored ICP and RANSAC are out of scope.
"""
import numpy as np

from .. import engine
from .pointcloud import PointCloud, as_points


def icp_registration(src_pcd, tgt_pcd, initial_transform, threshold=0.1, icp_type='point2point', search='f64',
                     p2plane='open3d', gicp_epsilon=float(False)):
    """Point-to-point / point-to-plane / generalized ICP with Open3D's ICPConvergenceCriteria(1e-6, 1e-6, 30)
    (utils/o3d_tools.py:46-50).  Accepts Open3D clouds, `PointCloud`, numpy arrays or torch tensors.

    Returns the reference's dict: fitness, inlier_rmse, correspondence_set (m,2) int, est_transform (4,4) float64,
    src_corr_pts, tgt_corr_pts.  Like the reference it estimates normals on both clouds first
    (utils/o3d_tools.py:29-30); they only influence the result for 'point2plane'.
    `search` selects the nearest-neighbour arithmetic ('f64' = the reference's double precision).
    'point2plane' steps follow Open3D's own semantics (F4L_ICP_P2PL_OPEN3D: Eigen's pivoted L D L^T, applied whenever there is a
    correspondence), not the batched calls' robust default; `p2plane='robust'` asks for that one.  (The solve's frame is the
    shifted one below: for a SINGULAR system -- whose "solution" depends on the frame -- that is not Open3D's number either.)
    'generalized_icp' (:40-41, 51-56) is `registration_generalized_icp` with the covariances Open3D derives from the normals of
    :29-30.  The reference builds its estimator as `TransformationEstimationForGeneralizedICP(False)`: the first parameter is
    `epsilon`, so it runs with epsilon = float(False) = 0.0 (plane-to-plane with nothing along the normals; Open3D's default is
    1e-3) -- the default here too; `gicp_epsilon` sets another."""
    import torch
    if icp_type not in ('point2point', 'point2plane', 'generalized_icp'):
        raise ValueError('ICP type not supported')  # utils/o3d_tools.py:43,58
    src = as_points(src_pcd)
    tgt = as_points(tgt_pcd)
    if hasattr(initial_transform, "detach"):
        initial_transform = initial_transform.detach().cpu().numpy()
    T0 = np.asarray(initial_transform, dtype=np.float64).reshape(4, 4).copy()
    dev = torch.device("cuda")
    # Open3D clouds are float64; the kernel's clouds are float32.  Georeferenced coordinates (~1e6 m: float32 spacing 0.06-
    # 0.25 m, above the 0.1 m correspondence distance) must not be cast as they are: both clouds are first moved, in double, by
    # an origin near the target (whole metres, so clouds that came from float32 keep their exact values), the transform is
    # carried along (q - o = R (p - o) + t'  <=>  t' = t - o + R o) and carried back afterwards.
    o = np.floor(np.asarray(tgt if len(tgt) else src, dtype=np.float64).min(axis=0)) if (len(tgt) or len(src)) else np.zeros(3)
    T0[:3, 3] = T0[:3, 3] - o + T0[:3, :3] @ o
    s = torch.from_numpy((np.asarray(src, dtype=np.float64) - o).astype(np.float32)).to(dev)
    t = torch.from_numpy((np.asarray(tgt, dtype=np.float64) - o).astype(np.float32)).to(dev)
    so = torch.tensor([0, s.shape[0]], dtype=torch.int64, device=dev)
    to = torch.tensor([0, t.shape[0]], dtype=torch.int64, device=dev)
    tn = sn = None
    if icp_type != 'point2point' or isinstance(tgt_pcd, PointCloud):
        tn = engine.patch_normals(t, to, 30, f64=True)  # (doubles, like the normals Open3D keeps)
        if isinstance(tgt_pcd, PointCloud):  # the reference mutates its inputs the same way
            tgt_pcd.normals = tn.cpu().numpy()
    if icp_type == 'generalized_icp' or isinstance(src_pcd, PointCloud):
        sn = engine.patch_normals(s, so, 30, f64=True)
        if isinstance(src_pcd, PointCloud):
            src_pcd.normals = sn.cpu().numpy()
    out = engine.piecewise_icp(s, so, t, to, init_T=torch.from_numpy(T0[None]).to(dev), max_corr_dist=threshold,
                               max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, icp_type=icp_type,
                               tgt_normals=tn if icp_type != 'point2point' else None, return_corr=True, search=search,
                               p2plane=p2plane, src_normals=sn if icp_type == 'generalized_icp' else None,
                               gicp_epsilon=gicp_epsilon)
    T = out["T"][0].cpu().numpy()
    T[:3, 3] = T[:3, 3] + o - T[:3, :3] @ o
    if int(out["iters"][0].item()) == -2:
        # a step of Open3D's own semantics was not finite (generalized ICP at the reference's epsilon = 0 on a pair of exactly
        # parallel normals): Open3D returns a NaN transform there; the kernel stopped at its last finite one (include/f4l.h)
        import warnings
        warnings.warn(f"icp_registration({icp_type!r}): a step was not finite (singular pair covariance?); Open3D would return "
                      "NaN -- the transform returned is the last finite one", RuntimeWarning, stacklevel=2)
    corr = out["corr"].cpu().numpy()
    sel = np.nonzero(corr >= 0)[0]
    corr_set = np.stack([sel, corr[sel]], axis=1).astype(np.int32) if len(sel) else np.zeros((0, 2), np.int32)
    return {
        "fitness": float(out["fitness"][0].item()),
        "inlier_rmse": float(out["rmse"][0].item()),
        "correspondence_set": corr_set,
        "est_transform": T,
        "src_corr_pts": src[corr_set[:, 0]],
        "tgt_corr_pts": tgt[corr_set[:, 1]],
    }


def get_correspondence_pairwise_point_clouds(src_key_pcd, tgt_key_pcd, corr_indices_set):
    """utils/o3d_tools.py:131-145"""
    src_key_pts, tgt_key_pts = as_points(src_key_pcd), as_points(tgt_key_pcd)
    corr = np.asarray(corr_indices_set)
    return src_key_pts[corr[:, 0], :], tgt_key_pts[corr[:, 1], :]


def array2pcd(point_array, colors=None, normals=None):
    """utils/o3d_tools.py:180-195"""
    return PointCloud(point_array, colors, normals)


def array2tensor(array, invert=False):
    """utils/o3d_tools.py:228-238"""
    import torch
    if invert:
        return np.asarray(array.cpu())
    if array.dtype == np.uint64:
        array = array.astype(np.int64)
    return torch.from_numpy(array)


def tensor2pcd(point_tensor, colors=None):
    """utils/o3d_tools.py:198-210"""
    pts = array2tensor(point_tensor, invert=True)
    cols = None if colors is None else array2tensor(colors, invert=True)
    return array2pcd(pts, colors=cols)


def pcd2array(point_cloud, return_colors=False):
    """utils/o3d_tools.py:213-225"""
    pts = np.array(as_points(point_cloud))
    if return_colors:
        return pts, np.array(point_cloud.colors)
    return pts


def pcd2tensor(point_cloud, device='cuda', return_colors=False):
    """utils/o3d_tools.py:241-257 (casts to float32 like the reference)"""
    import torch
    pts = torch.from_numpy(np.array(as_points(point_cloud))).float()
    if return_colors:
        return pts.to(device), torch.from_numpy(np.array(point_cloud.colors)).to(device)
    return pts.to(device)
