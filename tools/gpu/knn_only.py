import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
for _ in range(3): engine.knn(xyz, 30)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); engine.knn(xyz, 30); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"f4l_knn {n} points k=30:" +" min %.3f ms median %.3f ms" % (min(ts), sorted(ts)[5]))
if len(sys.argv) > 2 and sys.argv[2] == "knn":  # (the counter passes behind bench.py's roofline_knn: f4l_knn only, 13 calls)
    sys.exit(0)
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); engine.knn_normals(xyz, 30); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"f4l_knn_normals {n} points k=30:" + " min %.3f ms median %.3f ms" % (min(ts), sorted(ts)[5]))
