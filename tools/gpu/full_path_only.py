"""The whole hot path of one synthetic tile (pipeline.full_path), a few times: stage timings, and the program rocprofv3 is
pointed at for the per-kernel view of the stages around the patch loop.  Usage: full_path_only.py [n_points] [repeats]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import pipeline, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
c = synthetic.make_patches_device(n, int(round((n / 1e6) ** 0.5 * 45)), 1.386, dev, seed=0)
src, tgt = c["src"], c["tgt"]
del c
pipeline.full_path(src, tgt, max_iter=20, fixed_iters=True)
for _ in range(reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = pipeline.full_path(src, tgt, max_iter=20, fixed_iters=True)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t)
    print(f"wall {wall:.2f} ms  " + "  ".join(f"{k} {v:.2f}" for k, v in r["stage_ms"].items())
          + f"  [reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB]", flush=True)
print("K", int(r["K"]), "mean fitness", float(r["fitness"].mean()))
