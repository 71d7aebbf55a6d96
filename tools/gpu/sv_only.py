"""f4l_supervoxel_segment_device of a 1 M-point tile, a few times (the program the rocprofv3 passes of the segmentation's kernels
are pointed at).  Usage: sv_only.py [n_points] [resolution] [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
res = float(sys.argv[2]) if len(sys.argv) > 2 else 1.386
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
knn, nrm = engine.knn_normals(xyz, 30)
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); labels, info = engine.supervoxel_segment_device(xyz, nrm, knn, res); b.record(); torch.cuda.synchronize()
    i = info.cpu().numpy()
    print(f"n={n} res={res}: segmentation {a.elapsed_time(b):.2f} ms K={i[0]} status={i[2]} sweeps={i[3]} rounds={i[6]} cut={i[7]}", flush=True)
