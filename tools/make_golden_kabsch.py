#!/usr/bin/env python3
"""Generate tests/golden/kabsch_golden.npz by IMPORTING the reference's own torch functions.

Runs only in the build container (needs /root/reference).  open3d / cpp_core.pcd_tiling / sklearn
submodules that the reference imports at module scope but does not use on this path are replaced by
empty stub modules in sys.modules before import; no reference source is copied.

Pinned functions:
  scripts/weighted_svd.py:58-129   weighted_procrustes        (fp32 + fp64, weights, eps, thresholds)
  scripts/weighted_svd.py:10-55    weighted_svd (CPU part)     -- skipped: needs .cuda()
  src/functions.py:12-85           kabsch_transformation_estimation

Output: tests/golden/kabsch_golden.npz (inputs + expected outputs; data only).
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("F4L_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kabsch_golden.npz")


def _stub(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def import_reference():
    for name in ["open3d", "matplotlib", "matplotlib.pyplot", "cpp_core", "cpp_core.pcd_tiling",
                 "cpp_core.pcd_tiling.build", "cpp_core.pcd_tiling.build.pcd_tiling"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    sys.modules["cpp_core.pcd_tiling.build"].pcd_tiling = sys.modules["cpp_core.pcd_tiling.build.pcd_tiling"]
    sys.path.insert(0, REF)
    from scripts.weighted_svd import weighted_procrustes  # noqa
    from src.functions import kabsch_transformation_estimation  # noqa
    return weighted_procrustes, kabsch_transformation_estimation


def rot(rng, max_deg=180.0):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(rng.uniform(-max_deg, max_deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def main():
    wp, kabsch2 = import_reference()
    rng = np.random.default_rng(20240607)
    cases = {}
    ci = 0

    def add(name, src, tgt, w, eps, thr, dtype):
        nonlocal ci
        ts = torch.from_numpy(src.astype(dtype))
        tt = torch.from_numpy(tgt.astype(dtype))
        tw = None if w is None else torch.from_numpy(w.astype(dtype))
        R, t = wp(ts, tt, weights=tw, weight_thresh=thr, eps=eps, return_transform=False)
        key = f"c{ci:02d}_{name}"
        cases[key + "_src"] = src.astype(dtype)
        cases[key + "_tgt"] = tgt.astype(dtype)
        if w is not None:
            cases[key + "_w"] = w.astype(dtype)
        cases[key + "_eps"] = np.float64(eps)
        cases[key + "_thr"] = np.float64(thr)
        cases[key + "_R"] = R.numpy()
        cases[key + "_t"] = t.numpy()
        ci += 1

    for dtype in (np.float64, np.float32):
        for n in (3, 4, 10, 57, 500):
            R0, t0 = rot(rng), rng.uniform(-2, 2, 3)
            src = rng.uniform(-1, 1, (n, 3))
            tgt = src @ R0.T + t0 + rng.normal(0, 0.01, (n, 3))
            add(f"n{n}_plain", src, tgt, None, 1e-7, 0.0, dtype)
            w = rng.uniform(0, 1, n)
            add(f"n{n}_w", src, tgt, w, 1e-6, 0.0, dtype)
            add(f"n{n}_wthr", src, tgt, w, 1e-6, 0.3, dtype)
        # batched (B,N,3)
        B, n = 4, 33
        src = rng.uniform(-1, 1, (B, n, 3))
        tgt = np.stack([src[b] @ rot(rng).T + rng.uniform(-1, 1, 3) for b in range(B)]) + rng.normal(0, 0.005, (B, n, 3))
        add("batched", src, tgt, rng.uniform(0.1, 1, (B, n)), 1e-7, 0.0, dtype)
        # reflection case: target is a mirrored copy -> det(V U^T) < 0 branch (:111)
        src = rng.uniform(-1, 1, (40, 3))
        tgt = src * np.array([1.0, 1.0, -1.0]) + 0.3
        add("reflection", src, tgt, None, 1e-7, 0.0, dtype)
        # georeferenced magnitude (configs/landslide/fusion_brienz.yaml:6): large offsets, fp64 only meaningful
        src = rng.uniform(-1, 1, (100, 3)) + np.array([2647000.0, 1177000.0, 1500.0]) * (1.0 if dtype == np.float64 else 1e-3)
        R0 = rot(rng, 2.0)
        c = src.mean(0)
        tgt = (src - c) @ R0.T + c + np.array([0.05, -0.02, 0.01])
        add("georef", src, tgt, None, 1e-6, 0.0, dtype)
        # coplanar points (rank-2 H) -- still unique
        src = np.c_[rng.uniform(-1, 1, (30, 2)), np.zeros(30)]
        R0, t0 = rot(rng), rng.uniform(-1, 1, 3)
        tgt = src @ R0.T + t0
        add("coplanar", src, tgt, None, 1e-7, 0.0, dtype)

    # Kabsch #2 (src/functions.py:12-85), fp64
    for j, n in enumerate((5, 64)):
        B = 3
        x1 = rng.uniform(-1, 1, (B, n, 3))
        x2 = np.stack([x1[b] @ rot(rng).T + rng.uniform(-1, 1, 3) for b in range(B)]) + rng.normal(0, 0.01, (B, n, 3))
        w = rng.uniform(0, 1, (B, n))
        R, t, res, flag = kabsch2(torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(w.copy()))
        cases[f"k2_{j}_x1"], cases[f"k2_{j}_x2"], cases[f"k2_{j}_w"] = x1, x2, w
        cases[f"k2_{j}_R"], cases[f"k2_{j}_t"], cases[f"k2_{j}_res"] = R.numpy(), t.numpy(), res.numpy()
        R, t, res, flag = kabsch2(torch.from_numpy(x1), torch.from_numpy(x2), None)
        cases[f"k2_{j}_R_now"], cases[f"k2_{j}_t_now"] = R.numpy(), t.numpy()

    np.savez_compressed(OUT, **cases)
    print("wrote", os.path.normpath(OUT), len(cases), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
