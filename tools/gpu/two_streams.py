"""Throughput of the bench step when consecutive tiles are issued on alternating streams (the tail of one launch overlaps
the head of the next), against one stream."""
import os, sys, time
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
def step():
    return engine.patch_loop(src, so, tgt, to, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=0.1, max_iter=20, fixed_iters=True,
                             max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
for nstreams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 30
        for i in range(K):
            with torch.cuda.stream(streams[i % nstreams]):
                out = step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {1e3 * dt / K:.4f} ms per tile, {K * d['src'].shape[0] / dt / 1e6:.0f} Mpts/s")
