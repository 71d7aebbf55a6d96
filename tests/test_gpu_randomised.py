"""GPU suite: a short run of every randomised checker under tools/gpu/fuzz_*.py (the long runs are recorded in DESIGN.md section 4).
Each is a program of its own -- `python tools/gpu/fuzz_<what>.py <cases> <seed>` reproduces a case from its number -- and is started
here as a child process; the last line it prints must be FUZZ CLEAN."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [
    ("fuzz_icp.py", 30, 11),            # f4l_piecewise_icp against the C oracle: random patch sets, both estimators
    ("fuzz_gicp.py", 20, 4800000),     # f4l_piecewise_gicp against the oracle's restatement of Open3D's generalized estimator
    ("fuzz_knn.py", 30, 12),            # the kNN lane kernel against the wave-per-query search and a KD-tree
    ("fuzz_supervoxel.py", 20, 13),     # the device segmentation against its numpy model, label for label
    ("fuzz_ops.py", 40, 14),            # nn_query, voxel filter, ragged Kabsch, CSR, median, rigidity check
    ("fuzz_fine_matching.py", 20, 15),  # the batched loop body against the patch-by-patch replay of the reference's loop
    ("fuzz_full_path.py", 30, 16),      # the whole path on small clouds of random shape and overlap
    ("fuzz_piecewise_octree.py", 12, 17),  # the Piecewise_ICP entry (reference mode) against the pointer-octree restatement
])
def test_randomised_checker_is_clean(tool, cases, seed):
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu", tool), str(cases), str(seed)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert r.returncode == 0 and lines and lines[-1].startswith("FUZZ CLEAN"), (r.stdout[-2000:], r.stderr[-2000:])
