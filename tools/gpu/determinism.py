"""Run the fused loop repeatedly on the same input: outputs must be bit-identical run to run (no race, no order
dependence on scheduling), in both search modes and for every workgroup shape."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
d = synthetic.make_patches(300_000, 25, 1.386, seed=3)
dev = torch.device("cuda")
src, tgt = torch.from_numpy(d["src"]).to(dev), torch.from_numpy(d["tgt"]).to(dev)
so, to = torch.from_numpy(d["src_off"]).to(dev), torch.from_numpy(d["tgt_off"]).to(dev)
P = d["P"]
eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
nn, _ = engine.nn_refine(src, so, tgt, to, eye, torch.full((P,), 0.2, dtype=torch.float64, device=dev), return_rows=False)
cs, ct, coff = synthetic.correspondences_from_nn(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn.cpu().numpy())
cs, ct, coff = torch.from_numpy(cs).to(dev), torch.from_numpy(ct).to(dev), torch.from_numpy(coff).to(dev)
bad = 0
for waves in ("4", "2", "1"):
    os.environ["F4L_ICP_WAVES"] = waves
    for search in ("f32", "f64"):
        ref = None
        for rep in range(12):
            out = engine.patch_loop(src, so, tgt, to, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=0.1, max_iter=20,
                                    fixed_iters=(rep % 2 == 0), search=search, return_corr=True)
            key = (rep % 2 == 0)
            sig = (out["T"].clone(), out["fitness"].clone(), out["rmse"].clone(), out["iters"].clone(), out["corr"].clone(), out["rows"].clone())
            if ref is None: ref = {}
            if key not in ref: ref[key] = sig
            else:
                same = all(torch.equal(a, b) for a, b in zip(sig, ref[key]))
                if not same:
                    bad += 1
                    dT = (sig[0] - ref[key][0]).abs().max().item()
                    print(f"MISMATCH waves={waves} search={search} rep={rep} max|dT|={dT:.3e} corr diff={(sig[4]!=ref[key][4]).sum().item()}")
        print(f"waves={waves} search={search}: ok" if bad == 0 else f"waves={waves} search={search}: {bad} mismatches so far")
print("DETERMINISTIC" if bad == 0 else f"NONDETERMINISTIC: {bad}")
