import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
c = synthetic.two_epoch_cloud(1_000_000, 45, 1.386, seed=0)  # numpy: a few seconds
xyz = torch.from_numpy(c["src"]).cuda()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), r
ms, idx = t(lambda: engine.knn(xyz, 30))
print(f"knn30 1M: {ms:.2f} ms  -> {1e-3/ms*1e3:.1f} Mpts/s, {132e6/ms*1e3/1e9:.1f} GB/s algorithmic (132 B/pt)")
ms2, nrm = t(lambda: engine.normals(xyz, idx))
print(f"normals 1M: {ms2:.2f} ms")
ms3, _ = t(lambda: engine.knn_normals(xyz, 30))
print(f"knn30 + normals fused 1M: {ms3:.2f} ms")
os.environ["F4L_KNN_WAVE_PER_QUERY"] = "1"
ms4, _ = t(lambda: engine.knn(xyz, 30))
del os.environ["F4L_KNN_WAVE_PER_QUERY"]
print(f"knn30 1M, wave-per-query search (round 1): {ms4:.2f} ms")
t0 = time.perf_counter(); labels, K = engine.supervoxel(xyz, 30, 1.386); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"supervoxel end-to-end 1M (kNN+normals GPU, segmentation host): {t1-t0:.2f} s, K={K}")
