import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic, pipeline
for n, res in ((10_000_000, 0.52), (10_000_000, 1.386), (3_000_000, 0.9)):
    d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
    xyz = d["src"]
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); lab, K = engine.supervoxel(xyz, 30, res); torch.cuda.synchronize()
        print(f"f4l_supervoxel n={n} res={res}: {1e3 * (time.perf_counter() - t):.1f} ms K={K}  peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
    del d, xyz, lab
    torch.cuda.empty_cache()
d = synthetic.make_patches_device(1_000_000, 45, 1.386, torch.device("cuda"), seed=0)
for part in ("identical", "parallel"):
    for _ in range(2):
        r = pipeline.full_path(d["src"], d["tgt"], max_iter=20, fixed_iters=True, partition=part)
    print(part, "full path 1M:", "  ".join(f"{k} {v:.2f}" for k, v in r["stage_ms"].items()), "K", r["K"], flush=True)
