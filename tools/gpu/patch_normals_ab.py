"""f4l_patch_normals_f64 under two builds of the library, bit for bit: the product against a variant (tools/build_variant.sh), on the
bench's C2 cloud and on patches of awkward sizes (fewer points than k, a lattice, duplicates, a line, georeferenced coordinates).
Usage: patch_normals_ab.py <variant> [k ...]      (GPU box, repo root)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def compute(out):
    import torch
    from fusion4landslide_amd import engine, synthetic
    res = {}
    d = synthetic.make_patches_device(1_000_000, 45, 1.386, torch.device("cuda"), seed=0)
    rng = np.random.default_rng(9)
    parts = []
    for n in (0, 1, 2, 5, 29, 30, 31, 64, 257, 500, 1500, 4000, 8190):
        xy = rng.uniform(0, 1, (n, 2)) * max(n, 1) ** 0.5 * 0.05
        parts.append(np.c_[xy, 0.1 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1]) + rng.normal(0, 0.002, n)])
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3), indexing="ij"), -1).reshape(-1, 3) * 0.05
    parts += [g, np.repeat(parts[9][:120], 3, axis=0), np.c_[np.linspace(0, 1, 200), np.zeros(200), np.zeros(200)],
              parts[9] + np.array([2.6e6, 1.2e6, 1800.0]), np.c_[rng.uniform(0, 30, (3000, 1)), rng.uniform(0, 0.5, (3000, 2))]]
    pts = torch.from_numpy(np.concatenate(parts).astype(np.float32)).cuda()
    off = torch.from_numpy(np.concatenate([[0], np.cumsum([len(a) for a in parts])]).astype(np.int64)).cuda()
    for k in [int(a) for a in sys.argv[3:]] or [30, 8, 36]:
        res[f"C2_k{k}"] = engine.patch_normals(d["tgt"], d["tgt_off"], k, max_patch=d["max_tgt"], f64=True).cpu().numpy()
        res[f"odd_k{k}"] = engine.patch_normals(pts, off, k, f64=True).cpu().numpy()
    np.savez(out, **res)

if len(sys.argv) > 2 and sys.argv[1] == "--compute":
    compute(sys.argv[2]); sys.exit(0)
variant = sys.argv[1]
outs = []
for name, lib in (("product", ""), (variant, os.path.join(ROOT, "fusion4landslide_amd", "lib", "variants", f"lib_{variant}.so"))):
    out = f"/tmp/pn_{name}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--compute", out] + sys.argv[2:], check=True, env=dict(os.environ, F4L_LIB_PATH=lib))
    outs.append(np.load(out))
bad = 0
for key in outs[0].files:
    a, b = outs[0][key], outs[1][key]
    same = np.array_equal(a, b)
    bad += not same
    print(f"{key}: {a.shape[0]} normals, product == {variant} bit for bit: {same}" + ("" if same else f"  ({int((a != b).any(axis=1).sum())} differ, max |d| {np.abs(a - b).max():.3g})"), flush=True)
sys.exit(1 if bad else 0)
