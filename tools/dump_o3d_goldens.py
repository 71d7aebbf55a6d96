#!/usr/bin/env python3
"""Produce tests/golden/o3d_icp_golden.npz with a real Open3D (0.19.0 is what the reference pins, requirements.txt:1).

Open3D is not installable in the build container, so the ICP oracle (oracle/f4l_oracle.c) is pinned by known-answer
tests and an independent numpy restatement only.  Anyone with Open3D can run

    python tools/dump_o3d_goldens.py

from the repository root; tests/test_oracle_icp.py::test_icp_oracle_vs_open3d_goldens_when_present and the GPU parity
tests then compare against Open3D's own numbers.  The calls are the ones utils/o3d_tools.py:12-71 makes:
estimate_normals() with default parameters on both clouds, registration_icp(source, target, threshold, init,
estimator, ICPConvergenceCriteria(1e-6, 1e-6, 30)) with TransformationEstimationPointToPoint(False) or
TransformationEstimationPointToPlane(), and registration_generalized_icp with TransformationEstimationForGeneralizedICP(False)
-- the reference's own spelling, :41; `epsilon_` of that estimator is stored too, so that the file says what `False` meant --
and with Open3D's default epsilon.  Inputs are seeded numpy data (no files needed).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import open3d as o3d
    from _util import rot_from_axis_angle

    rng = np.random.default_rng(2024)
    threshold = 0.1
    out = dict(threshold=np.float64(threshold), open3d_version=np.array(o3d.__version__))
    cases = []
    for c, (n_tgt, n_src, angle, shift, noise) in enumerate([(900, 600, 0.004, 0.02, 0.002), (2500, 2000, 0.01, 0.03, 0.005),
                                                             (300, 40, 0.002, 0.01, 0.0), (5000, 4000, 0.006, 0.05, 0.003)]):
        xy = rng.uniform(0, 2.0, (n_tgt, 2))
        surf = lambda q: 0.25 * np.sin(2.1 * q[:, 0]) * np.cos(1.7 * q[:, 1]) + 0.05 * np.sin(9 * q[:, 0] + 1) * np.sin(7 * q[:, 1])
        tgt = np.c_[xy, surf(xy) + rng.normal(0, noise, n_tgt)]
        xy2 = rng.uniform(0.15, 1.85, (n_src, 2))
        R0 = rot_from_axis_angle(rng.normal(size=3), angle)
        src = np.c_[xy2, surf(xy2)] @ R0.T + rng.uniform(-shift, shift, 3)
        # float32 storage, like the reference's clouds (pcd2tensor, utils/o3d_tools.py:241-257)
        src, tgt = src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64)
        init = np.eye(4)
        cases.append((src, tgt, init))
        out[f"src_{c}"], out[f"tgt_{c}"], out[f"init_{c}"] = src, tgt, init
        for icp_type in ("point2point", "point2plane"):
            s, t = o3d.geometry.PointCloud(), o3d.geometry.PointCloud()
            s.points, t.points = o3d.utility.Vector3dVector(src), o3d.utility.Vector3dVector(tgt)
            s.estimate_normals()
            t.estimate_normals()
            est = (o3d.pipelines.registration.TransformationEstimationPointToPoint(False) if icp_type == "point2point"
                   else o3d.pipelines.registration.TransformationEstimationPointToPlane())
            reg = o3d.pipelines.registration.registration_icp(
                s, t, threshold, init, est,
                o3d.pipelines.registration.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30))
            out[f"T_{icp_type}_{c}"] = np.asarray(reg.transformation, dtype=np.float64)
            out[f"fitness_{icp_type}_{c}"] = np.float64(reg.fitness)
            out[f"rmse_{icp_type}_{c}"] = np.float64(reg.inlier_rmse)
            out[f"corr_{icp_type}_{c}"] = np.asarray(reg.correspondence_set, dtype=np.int32)
            if icp_type == "point2plane":
                out[f"tgt_normals_{c}"] = np.asarray(t.normals, dtype=np.float64)
                out[f"src_normals_{c}"] = np.asarray(s.normals, dtype=np.float64)
        for tag, est in (("generalized_icp", o3d.pipelines.registration.TransformationEstimationForGeneralizedICP(False)),
                         ("generalized_icp_default", o3d.pipelines.registration.TransformationEstimationForGeneralizedICP())):
            s, t = o3d.geometry.PointCloud(), o3d.geometry.PointCloud()
            s.points, t.points = o3d.utility.Vector3dVector(src), o3d.utility.Vector3dVector(tgt)
            s.estimate_normals()
            t.estimate_normals()
            reg = o3d.pipelines.registration.registration_generalized_icp(
                s, t, threshold, init, est,
                o3d.pipelines.registration.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30))
            out[f"epsilon_{tag}"] = np.float64(est.epsilon)
            out[f"T_{tag}_{c}"] = np.asarray(reg.transformation, dtype=np.float64)
            out[f"fitness_{tag}_{c}"] = np.float64(reg.fitness)
            out[f"rmse_{tag}_{c}"] = np.float64(reg.inlier_rmse)
    out["n_cases"] = np.int64(len(cases))
    path = os.path.join(ROOT, "tests", "golden", "o3d_icp_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(cases), "cases, Open3D", o3d.__version__)


if __name__ == "__main__":
    main()
