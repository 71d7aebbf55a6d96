"""Is the fused loop capturable into a HIP graph (torch.cuda.CUDAGraph)?  Single launch and size-class launches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
dev = torch.device("cuda")
def uneven(seed=2, P=700):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(10, 300, P)
    src_l, tgt_l = [], []
    for m in sizes:
        side = max(0.2, np.sqrt(m / 400.0))
        xy = rng.uniform(0, side, (m, 2))
        t = np.c_[xy, 0.3 * np.sin(1.7 * xy[:, 0]) * np.cos(2.3 * xy[:, 1])]
        tgt_l.append(t)
        src_l.append(t + rng.uniform(-0.02, 0.02, 3) + rng.normal(0, 0.002, t.shape))
    off = np.zeros(P + 1, np.int64); np.cumsum(sizes, out=off[1:])
    return dict(src=np.concatenate(src_l).astype(np.float32), tgt=np.concatenate(tgt_l).astype(np.float32), src_off=off, tgt_off=off.copy(),
                P=P, max_src=int(sizes.max()), max_tgt=int(sizes.max()))


for name, spec in {"C1-like even patches": (50_000, 8, 1.386), "uneven patches (size classes, helper streams)": None}.items():
    d = synthetic.make_patches(*spec, seed=1) if spec else uneven()
    src, tgt = torch.from_numpy(d["src"]).to(dev), torch.from_numpy(d["tgt"]).to(dev)
    so, to = torch.from_numpy(d["src_off"]).to(dev), torch.from_numpy(d["tgt_off"]).to(dev)
    kw = dict(max_corr_dist=0.1, max_iter=20, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
    ref = engine.piecewise_icp(src, so, tgt, to, **kw)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            engine.piecewise_icp(src, so, tgt, to, **kw)  # warm-up on the side stream
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = engine.piecewise_icp(src, so, tgt, to, **kw)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out["T"], ref["T"])
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20): engine.piecewise_icp(src, so, tgt, to, **kw)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
        print(f"{name}: P={d['P']} max_src={d['max_src']} captured OK, identical={same}, replay {1e3*tg:.3f} ms vs eager {1e3*te:.3f} ms")
    except Exception as e:
        print(f"{name}: capture FAILED: {type(e).__name__}: {str(e)[:300]}")
