"""Kernel time of f4l_piecewise_icp versus the number of patches launched (first K patches of the C2 tile)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
d = synthetic.make_patches(1_000_000, 45, 1.386, seed=0)
dev = torch.device("cuda")
src, tgt = torch.from_numpy(d["src"]).to(dev), torch.from_numpy(d["tgt"]).to(dev)
so_h, to_h = d["src_off"], d["tgt_off"]
for K in [int(x) for x in sys.argv[1:]] or [64, 128, 256, 512, 768, 1024, 1536, 2025]:
    so, to = torch.from_numpy(so_h[:K + 1].copy()).to(dev), torch.from_numpy(to_h[:K + 1].copy()).to(dev)
    ts = []
    for it in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = engine.piecewise_icp(src, so, tgt, to, max_corr_dist=0.1, max_iter=20, fixed_iters=True,
                                   max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"K={K:5d}  kernel {min(ts[2:])*1e3:8.1f} us   per patch-slot {min(ts[2:])*1e3/max(1,-(-K//256)):7.1f} us/(WG per CU)")
