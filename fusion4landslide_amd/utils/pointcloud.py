"""A minimal stand-in for the two Open3D types the hot path touches, so that call sites written against
`o3d.geometry.PointCloud` (utils/o3d_tools.py:180-257) keep working without Open3D: float64 numpy arrays in
`.points / .colors / .normals`.  It only stores data; all computation happens on the GPU through the C ABI."""
import numpy as np


class PointCloud:
    def __init__(self, points=None, colors=None, normals=None):
        self.points = np.zeros((0, 3)) if points is None else np.asarray(points, dtype=np.float64).reshape(-1, 3)
        self.colors = None if colors is None else np.asarray(colors, dtype=np.float64).reshape(-1, 3)
        self.normals = None if normals is None else np.asarray(normals, dtype=np.float64).reshape(-1, 3)

    def has_normals(self):
        return self.normals is not None and len(self.normals) == len(self.points)

    def estimate_normals(self, knn=30):
        """`pcd.estimate_normals()` (Open3D default KDTreeSearchParamKNN(30)); runs on the GPU."""
        import torch

        from .. import engine
        pts = torch.from_numpy(self.points.astype(np.float32)).cuda()
        off = torch.tensor([0, pts.shape[0]], dtype=torch.int64, device="cuda")
        self.normals = engine.patch_normals(pts, off, knn).cpu().numpy().astype(np.float64)
        return self

    def select_by_index(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        return PointCloud(self.points[idx], None if self.colors is None else self.colors[idx],
                          None if self.normals is None else self.normals[idx])

    def __len__(self):
        return len(self.points)


def as_points(obj):
    """(n,3) float64 numpy view of an Open3D cloud, our PointCloud, a numpy array or a torch tensor."""
    if hasattr(obj, "points"):
        return np.asarray(obj.points, dtype=np.float64).reshape(-1, 3)
    if hasattr(obj, "detach"):
        return obj.detach().cpu().numpy().astype(np.float64).reshape(-1, 3)
    return np.asarray(obj, dtype=np.float64).reshape(-1, 3)
