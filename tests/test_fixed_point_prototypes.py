"""CPU twin of tests/test_gpu_supervoxel_exact.py: the two host prototypes of the fixed-point formulation
(tools/experiments/fixed_point_fusion_proto.cpp, fixed_point_exchange_proto.cpp) -- the reference's sequential fusion
(supervoxel_segmentation.h:117-176) and its FIFO exchange (:186-237) as fixed points of synchronous parallel passes -- against a
plain sequential replay of the same sequence, every label, on small clouds in two index orders.  They are what
csrc/supervoxel_exact.hip was written from; this test keeps the ALGORITHM pinned where there is no GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def protos(tmp_path_factory):
    d = tmp_path_factory.mktemp("fp")
    out = {}
    for name in ("fixed_point_fusion_proto", "fixed_point_exchange_proto"):
        exe = str(d / name)
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "experiments", name + ".cpp"), "-o", exe])
        out[name] = exe
    return out, d


@pytest.mark.parametrize("order", ["random", "rows"])
def test_fixed_point_passes_reproduce_the_sequential_segmentation(protos, order):
    from scipy.spatial import cKDTree
    from oracle import oracle as O
    exes, d = protos
    rng = np.random.default_rng(7)
    n, k, res = 4000, 12, 1.0
    xy = rng.uniform(0, 12, (n, 2))
    p = np.c_[xy, 0.4 * np.sin(0.8 * xy[:, 0]) * np.cos(0.6 * xy[:, 1]) + rng.normal(0, 0.01, n)].astype(np.float32)
    if order == "rows":  # (rows along x, one after the other: long chains of index-ordered dependencies)
        p = np.ascontiguousarray(p[np.lexsort((p[:, 0], np.floor(p[:, 1] / 0.3)))])
    _, idx = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
    idx = idx.astype(np.int32)
    nrm = O.normals_from_knn(p, idx)
    case = str(d / f"case_{order}.bin")
    with open(case, "wb") as f:
        f.write(np.int32(n).tobytes()); f.write(np.int32(k).tobytes()); f.write(np.float64(res).tobytes())
        f.write(p.tobytes()); f.write(nrm.astype(np.float64).tobytes()); f.write(idx.tobytes())
    roots = str(d / f"roots_{order}.bin")
    r = subprocess.run([exes["fixed_point_fusion_proto"], case], capture_output=True, text=True, env=dict(os.environ, FPF_DUMP=roots), timeout=300)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "labels differing from the sequential replay: 0" in r.stdout
    r = subprocess.run([exes["fixed_point_exchange_proto"], case, roots], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "labels differing from the sequential FIFO: 0" in r.stdout
