"""f4l_patch_normals_f64 (Open3D's estimate_normals inside every patch, utils/o3d_tools.py:29-30) on the bench's clouds.
Usage: time_patch_normals.py [config ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
for name in (sys.argv[1:] or ["C4_50M_100k", "C2_1M_2k"]):
    c = synthetic.CONFIGS[name]
    d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], torch.device("cuda"), seed=0)
    f = lambda: engine.patch_normals(d["tgt"], d["tgt_off"], 30, max_patch=d["max_tgt"], f64=True)
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name}: f4l_patch_normals_f64 of {d['tgt'].shape[0]} target points in {d['P']} patches: {min(ts):.2f} ms", flush=True)
    del d
