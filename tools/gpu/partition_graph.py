"""f4l_supervoxel_parallel eager (grid sized with the host / on the device) against a HIP-graph replay of the same call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)

print(f"n={n} res={res:.4f}")
print("eager, grid sized with the host:   %.2f ms" % timed(lambda: engine.supervoxel_parallel(xyz, 30, res, read_count=False)))
os.environ["F4L_KNN_ASYNC"] = "1"
print("eager, grid sized on the device:   %.2f ms" % timed(lambda: engine.supervoxel_parallel(xyz, 30, res, read_count=False)))
print("   f4l_knn alone (device sizing):  %.2f ms" % timed(lambda: engine.knn(xyz, 30)))
del os.environ["F4L_KNN_ASYNC"]
print("   f4l_knn alone (host sizing):    %.2f ms" % timed(lambda: engine.knn(xyz, 30)))
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(s):
    engine.supervoxel_parallel(xyz, 30, res, read_count=False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        labels, info = engine.supervoxel_parallel(xyz, 30, res, read_count=False)
print("HIP graph replay:                  %.2f ms" % timed(lambda: g.replay()), " K =", int(info[0]))
