"""What one more pass costs: the fused loop body (f4l_patch_loop, float64 search, fixed iterations) of a config timed with
0, 1, 2, ... iterations -- the launch's fixed part (grid build, Kabsch start, pass 0's full search, the displacement rows) and the
price of pass k as the difference of two launches.
    python3 tools/gpu/icp_time_per_pass.py [config]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fusion4landslide_amd import engine, synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4_50M_100k"
dev = torch.device("cuda")
c = synthetic.CONFIGS[cfg]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev)
prob = bench.Problem(torch, engine, synthetic, d, dev)


def launch(mi):
    return engine.patch_loop(d["src"], d["src_off"], d["tgt"], d["tgt_off"], prob.cs, prob.ct, prob.coff, None, 0.0, 1e-6,
                             max_corr_dist=bench.MAX_CORR, max_iter=mi, fixed_iters=True, max_src_patch=d["max_src"],
                             max_tgt_patch=d["max_tgt"], search="f64")


prev = None
for mi in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 18, 20):
    for _ in range(3):
        launch(mi)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(mi); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    note = "" if prev is None else f"   {(t - prev[1]) / (mi - prev[0]):6.3f} ms per pass since {prev[0]}"
    print(f"{cfg}: {mi:2d} iterations {t:8.3f} ms{note}", flush=True)
    prev = (mi, t)
