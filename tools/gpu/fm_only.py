"""The batched loop body (fine_matching_3d) of 16 C2 tiles, repeated: the program the loop body's kernel stats are pointed at.

    python3 tools/gpu/fm_only.py [repeats] [check|nocheck|check32|all]
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
from fusion4landslide_amd.src.fine_matching import fine_matching_3d
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
what = sys.argv[2] if len(sys.argv) > 2 else "nocheck"
kw = dict(nocheck=dict(), check=dict(remove_low_quality_patch_matches=True), check32=dict(remove_low_quality_patch_matches=True, rigidity_precision="f32"),
          all=dict(remove_low_quality_patch_matches=True, assign_type="assign_then_nn", output_tgt2src=True, median_max_resolution=0.03))[what]
dev = torch.device("cuda")
c = synthetic.CONFIGS["C2x16_16M_32k"]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
src, soff, tgt, toff, P = d["src"], d["src_off"], d["tgt"], d["tgt_off"], d["P"]
n = src.shape[0]
eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
thr = torch.full((P,), 0.2, dtype=torch.float64, device=dev)
nn, _ = engine.nn_refine(src, soff, tgt, toff, eye, thr, max_tgt_patch=d["max_tgt"], return_rows=False)
pid = torch.repeat_interleave(torch.arange(P, device=dev), soff[1:] - soff[:-1])
corr = torch.where(nn >= 0, toff[pid] + nn.to(torch.int64), torch.full_like(pid, -1))
sid, tid = torch.arange(n, device=dev), torch.arange(tgt.shape[0], device=dev)
torch.cuda.synchronize()
print("FM_BEGIN", flush=True)
for _ in range(reps):
    t0 = time.perf_counter()
    r = fine_matching_3d(src, tgt, sid, soff, tid, toff, corr, thres_dist_diff=0.03, **kw)
    torch.cuda.synchronize()
    print(f"{what}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
