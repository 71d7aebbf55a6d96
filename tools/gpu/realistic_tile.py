"""A tile the way the reference cuts it: supervoxel partition of epoch 1 (irregular patches of very different sizes),
epoch-2 points assigned to the patch of their nearest epoch-1 point, then the per-patch Kabsch + ICP + rows loop and
the nearest-neighbour refinement.  Stage timings on one GPU (synthetic C2-density cloud)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cells = int(round((n / 1e6) ** 0.5 * 45))
c = synthetic.two_epoch_cloud(n, cells, 1.386, seed=0)
dev = torch.device("cuda")
src, tgt = torch.from_numpy(c["src"]).to(dev), torch.from_numpy(c["tgt"]).to(dev)


def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{name:38s} {1e3 * (time.perf_counter() - t):9.2f} ms", flush=True)
    return out


res = timed("median resolution (2-NN, both epochs)", lambda: engine.median_resolution(src, tgt))
resolution = max(np.sqrt(3.0) * 10.0 * res, 0.1)
labels, K = timed(f"supervoxel partition (res {resolution:.3f} m)", lambda: engine.supervoxel(src, 30, resolution))
order_s, off_s = timed("labels -> CSR (source)", lambda: engine.labels_to_csr(labels, K))
nn = timed("label transfer: 1-NN of epoch 2 in 1", lambda: engine.nn_query(src, tgt, 1))
tl = labels[nn[:, 0].long()]
order_t, off_t = timed("labels -> CSR (target)", lambda: engine.labels_to_csr(tl, K))
ps, pt = engine.gather_points(src, order_s), engine.gather_points(tgt, order_t)
sz_s, sz_t = (off_s[1:] - off_s[:-1]).cpu().numpy(), (off_t[1:] - off_t[:-1]).cpu().numpy()
print(f"K = {K} patches; source sizes min/median/max {sz_s.min()}/{int(np.median(sz_s))}/{sz_s.max()}, "
      f"target {sz_t.min()}/{int(np.median(sz_t))}/{sz_t.max()}")
for search in ("f64", "f32"):
    for it in range(3):
        out = timed(f"piecewise ICP, 20 fixed iters ({search})", lambda: engine.piecewise_icp(
            ps, off_s, pt, off_t, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search=search))
print("mean fitness", float(out["fitness"].mean()), " Mpts/s (last):", "see above")
thr = torch.clamp(2.0 * out["rmse"], min=res)
timed("nn_refine", lambda: engine.nn_refine(ps, off_s, pt, off_t, out["T"], thr))
os.environ["F4L_ICP_NOCLASSES"] = "1"
for search in ("f64", "f32"):
    for it in range(3):
        out2 = timed(f"piecewise ICP ({search}), one launch for all sizes", lambda: engine.piecewise_icp(
            ps, off_s, pt, off_t, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search=search))
print("same T:", float((out2["T"] - out["T"]).abs().max()))
