"""The reference's labels at the size of BASELINE.json configs[4]: f4l_supervoxel on the device (csrc/supervoxel_exact.hip) against the
one-core replay of the reference's sequence (csrc/supervoxel_host.cpp, F4L_SV_EXACT_HOST=1; pinned by the reference-compiled
fixtures) on the SAME n-point cloud -- every label.  The replay takes ~1.1 s per million points and ~30 GB of host memory at 100 M.
Usage: svx_100M_vs_host.py [n_points]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
del d
res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))
os.environ.pop("F4L_SV_EXACT_HOST", None)
os.environ["F4L_SV_EXACT_DEBUG"] = "1"
engine.supervoxel(xyz[:1_000_000].contiguous(), 30, res)  # warm-up
torch.cuda.synchronize(); t = time.perf_counter()
lab_d, K_d = engine.supervoxel(xyz, 30, res)
torch.cuda.synchronize(); t_dev = time.perf_counter() - t
os.environ.pop("F4L_SV_EXACT_DEBUG")
print(f"device: {n} points, resolution {res:.4f} m: K = {K_d}, {1e3 * t_dev:.1f} ms ({n / t_dev / 1e6:.1f} Mpts/s), peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
cnt = torch.bincount(lab_d.long(), minlength=K_d)
print(f"labels 0 .. {int(lab_d.max())}, smallest supervoxel {int(cnt.min())} points, largest {int(cnt.max())}", flush=True)
lab_d = lab_d.cpu()
engine.release_scratch(); torch.cuda.empty_cache()
os.environ["F4L_SV_EXACT_HOST"] = "1"
t = time.perf_counter()
lab_h, K_h = engine.supervoxel(xyz, 30, res)
torch.cuda.synchronize(); t_host = time.perf_counter() - t
diff = int((lab_h.cpu() != lab_d).sum())
print(f"host replay: K = {K_h}, {t_host:.1f} s; labels that differ: {diff} of {n}", flush=True)
sys.exit(0 if (diff == 0 and K_h == K_d) else 1)
