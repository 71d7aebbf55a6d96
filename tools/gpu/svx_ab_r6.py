"""A/B of the exact segmentation's narrow evaluation kernel (round 6): F4L_SV_EXACT_SCAN=1 = the linear visited scan of round 5 (on the packed
records), default = the LDS hash set.  Labels must be equal; times per call.  Usage: svx_ab_r6.py [n_points ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
    xyz = d["src"]
    res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))
    out = {}
    for mode in ("scan", "hash"):
        os.environ.pop("F4L_SV_EXACT_SCAN", None)
        if mode == "scan":
            os.environ["F4L_SV_EXACT_SCAN"] = "1"
        engine.supervoxel(xyz, 30, res)
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); labels, K = engine.supervoxel(xyz, 30, res); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        out[mode] = (labels, K)
        print(f"n={n} res={res:.4f} {mode}: {min(ts):.2f} ms (of {[round(t, 2) for t in ts]}), K = {K}", flush=True)
    os.environ.pop("F4L_SV_EXACT_SCAN", None)
    print("labels equal:", bool(torch.equal(out["scan"][0], out["hash"][0])) and out["scan"][1] == out["hash"][1], flush=True)
    if n <= 2_000_000:
        os.environ["F4L_SV_EXACT_HOST"] = "1"
        lab_h, K_h = engine.supervoxel(xyz, 30, res)
        os.environ.pop("F4L_SV_EXACT_HOST")
        print("equal to the host replay:", bool(torch.equal(lab_h, out["hash"][0])) and K_h == out["hash"][1], flush=True)
    del d, xyz, out
    engine.release_scratch(); torch.cuda.empty_cache()
