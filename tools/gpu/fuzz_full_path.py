"""Randomised run of the whole hot path (pipeline.full_path) on small two-epoch clouds of random size, shape and overlap: it must
finish (or refuse with a clear error), and what it returns must hang together -- labels 0..K-1 all used, the patch order a
permutation, rows = [s, T s] of every source point in patch order, orthonormal transforms, fitness in [0, 1], the sparse rows a
subset of the targets; both partitions.   python3 tools/gpu/fuzz_full_path.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import pipeline

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([40, 200, 2000, 15_000, 60_000]))
    m = max(31, int(n * rng.choice([0.3, 1.0, 1.7])))
    side = float(rng.choice([2.0, 10.0, 30.0]))
    kind = rng.choice(["surface", "rough", "gap", "shifted"])
    surf = lambda xy: 0.05 * side * np.sin(xy[:, 0] * 5 / side) * np.cos(xy[:, 1] * 3 / side)
    xy = rng.uniform(0, side, (n, 2)); src = np.c_[xy, surf(xy) + rng.normal(0, 0.002, n)]
    xy2 = rng.uniform(0, side, (m, 2))
    if kind == "gap": xy2[:, 0] = xy2[:, 0] * 0.5          # the second epoch covers half the tile
    tgt = np.c_[xy2, surf(xy2) + rng.normal(0, 0.002 if kind != "rough" else 0.05, m)]
    if kind == "shifted": tgt += rng.uniform(-0.3, 0.3, 3)
    off = np.array([2647.0, 1177.0, 1500.0]) if rng.random() < 0.3 else np.zeros(3)
    s32, t32 = (src + off).astype(np.float32), (tgt + off).astype(np.float32)
    part = "parallel" if rng.random() < 0.7 else "identical"
    thr = float(rng.choice([0.05, 0.1, 0.3]))
    try:
        r = pipeline.full_path(torch.from_numpy(s32).cuda(), torch.from_numpy(t32).cuda(), partition=part, icp_threshold=thr,
                               max_iter=int(rng.choice([5, 30])), fixed_iters=bool(rng.random() < 0.3))
    except (ValueError, RuntimeError) as e:
        print(f"case {seed0 + case:4d} {kind:8s} n={n:6d} m={m:6d} {part:9s} REFUSED: {type(e).__name__}: {str(e)[:120]}", flush=True)
        bad += 1
        continue
    K = r["K"]
    lab = r["labels"].cpu().numpy(); order = r["order"].cpu().numpy(); rows = r["rows"].cpu().numpy(); T = r["T"].cpu().numpy()
    so = r["src_off"].cpu().numpy(); fit = r["fitness"].cpu().numpy(); it = r["iters"].cpu().numpy()
    flags = {"labels": bool(lab.min() == 0 and lab.max() == K - 1 and len(np.unique(lab)) == K),
             "order": bool(np.array_equal(np.sort(order), np.arange(n)) and so[0] == 0 and so[-1] == n and (np.diff(so) > 0).all()),
             "rows src": bool(np.array_equal(rows[:, :3], s32[order]))}
    pid = np.repeat(np.arange(K), np.diff(so))
    s = s32[order].astype(np.float64)
    moved = np.einsum("nij,nj->ni", T[pid, :3, :3], s) + T[pid, :3, 3]
    flags["rows moved"] = bool(np.abs(moved - rows[:, 3:]).max() <= 2e-6 * max(1.0, np.abs(moved).max()) + 1e-6)
    R = T[:, :3, :3]
    flags["rotations"] = bool(np.abs(R @ np.transpose(R, (0, 2, 1)) - np.eye(3)).max() < 1e-6 and np.abs(np.linalg.det(R) - 1).max() < 1e-6)
    flags["fitness"] = bool((fit >= 0).all() and (fit <= 1).all() and np.isfinite(T).all() and (it >= 0).all())
    sp = r["sparse"].cpu().numpy()
    if len(sp):
        tset = {tuple(v) for v in t32.round(4).tolist()}
        flags["sparse"] = all(tuple(v) in tset for v in sp[:200, 3:].round(4).tolist())
    ok = all(flags.values())
    bad += not ok
    print(f"case {seed0 + case:4d} {kind:8s} n={n:6d} m={m:6d} {part:9s} r={thr:.2f} K={K:5d} mean fitness {fit.mean():.2f}  {'ok' if ok else 'MISMATCH ' + str([f for f, v in flags.items() if not v])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} PROBLEMS")
