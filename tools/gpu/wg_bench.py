"""Per-workgroup durations of one bench step (profiling build, F4L_ICP_PROF_WG) against per-patch properties."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
d = synthetic.make_patches(1_000_000, 45, 1.386, seed=0)
dev = torch.device("cuda")
src, tgt = torch.from_numpy(d["src"]).to(dev), torch.from_numpy(d["tgt"]).to(dev)
so, to = torch.from_numpy(d["src_off"]).to(dev), torch.from_numpy(d["tgt_off"]).to(dev)
P = d["P"]
eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
nn, _ = engine.nn_refine(src, so, tgt, to, eye, torch.full((P,), 0.2, dtype=torch.float64, device=dev), max_tgt_patch=d["max_tgt"], return_rows=False)
cs_h, ct_h, coff_h = synthetic.correspondences_from_nn(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn.cpu().numpy())
cs, ct, coff = torch.from_numpy(cs_h).to(dev), torch.from_numpy(ct_h).to(dev), torch.from_numpy(coff_h).to(dev)
os.environ["F4L_ICP_PROF"] = "1"
os.environ["F4L_ICP_PROF_WG"] = "/tmp/wg.bin"
for _ in range(3):
    out = engine.patch_loop(src, so, tgt, to, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=0.1, max_iter=20, fixed_iters=True,
                            max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], return_corr=True)
torch.cuda.synchronize()
wg = np.fromfile("/tmp/wg.bin", dtype=np.uint64).reshape(-1, 2).astype(np.int64)
dur = (wg[:, 1] - wg[:, 0]) * 0.01
start = (wg[:, 0] - wg[:, 0].min()) * 0.01
fit = out["fitness"].cpu().numpy()
ns, nt = np.diff(d["src_off"]), np.diff(d["tgt_off"])
ncorr = np.diff(coff_h)
print("dur percentiles 1/5/25/50/75/95/99/max:", np.percentile(dur, [1, 5, 25, 50, 75, 95, 99, 100]).round(1))
for name, v in (("ns", ns), ("nt", nt), ("fitness", fit), ("n_corr_init", ncorr), ("start", start)):
    print(f"corr(dur, {name}) = {np.corrcoef(dur, v)[0, 1]:.3f}")
for lo, hi in ((0, 0.05), (0.05, 0.5), (0.5, 0.9), (0.9, 1.01)):
    m = (fit >= lo) & (fit < hi)
    print(f"fitness [{lo}, {hi}): {m.sum()} patches, dur mean {dur[m].mean():.0f} us, max {dur[m].max():.0f}")
slow = np.argsort(dur)[-10:]
print("slowest:", [(int(p), round(float(dur[p])), int(ns[p]), int(nt[p]), round(float(fit[p]), 2)) for p in slow])
np.savez("gpurun_out/wg_stats.npz", dur=dur, start=start, fit=fit, ns=ns, nt=nt, ncorr=ncorr)
