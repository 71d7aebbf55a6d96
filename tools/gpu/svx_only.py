"""f4l_supervoxel (kNN + normals + the REFERENCE's segmentation on the device, csrc/supervoxel_exact.hip) of a synthetic tile at the
path's own resolution rule (sqrt(3) * 10 * median point spacing, src/coarse_to_fine_matching_base.py:2668-2671), CALLS times: the
program the rocprofv3 passes behind bench.py's `roofline_supervoxel` (the default partition) are pointed at.
Usage: svx_only.py [n_points] [calls]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))
for _ in range(calls):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); labels, K = engine.supervoxel(xyz, 30, res); b.record(); torch.cuda.synchronize()
    print(f"f4l_supervoxel {n} points, resolution {res:.4f}: {a.elapsed_time(b):.2f} ms, K = {K}", flush=True)
