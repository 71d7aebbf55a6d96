"""Randomised check of the reference's labels on the device (f4l_supervoxel -> csrc/supervoxel_exact.hip) against the one-core replay of
the reference's sequence (F4L_SV_EXACT_HOST=1, csrc/supervoxel_host.cpp: pinned by the reference-compiled fixtures) on clouds of
random size, shape, order, density, resolution and k -- every label -- and, with F4L_SV_EXACT_DEBUG, which evaluation shape ran
(narrow hash-set kernel, wide restart, host fall-back).   python3 tools/gpu/fuzz_supervoxel_exact.py [cases] [seed] [big]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine
from fusion4landslide_amd._lib import F4LError

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([300_000, 1_000_000] if big else [60, 400, 3000, 12_000, 40_000, 120_000]))
    k = int(rng.choice([4, 8, 16, 30, 40]))
    k = min(k, n - 1)
    kind = rng.choice(["surface", "rough", "volume", "two sheets", "strip", "lattice", "clumps"])
    side = float(rng.choice([3.0, 10.0, 40.0]))
    if kind == "surface":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, 0.1 * side * np.sin(xy[:, 0] * 6 / side) * np.cos(xy[:, 1] * 4 / side)]
    elif kind == "rough":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, rng.normal(0, 0.02 * side, n)]
    elif kind == "volume":
        p = rng.uniform(0, side, (n, 3)) * np.array([1, 1, 0.3])
    elif kind == "two sheets":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, np.where(rng.random(n) < 0.5, 0.0, 0.15 * side) + rng.normal(0, 0.002 * side, n)]
    elif kind == "strip":
        p = np.c_[rng.uniform(0, 8 * side, n), rng.uniform(0, 0.1 * side, n), rng.normal(0, 0.003 * side, n)]
    elif kind == "lattice":  # exactly equal distances, duplicated points
        m = max(2, int(round(n ** 0.5)))
        gx, gy = np.meshgrid(np.arange(m) * side / m, np.arange(m) * side / m)
        p = np.c_[gx.ravel(), gy.ravel(), np.zeros(m * m)]
        p = np.r_[p, p[rng.integers(0, len(p), max(1, len(p) // 50))]]
    else:  # clumps: dense blobs far apart (closures that differ wildly in size)
        c = rng.uniform(0, side, (max(2, n // 400), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01 * side, (n, 3)) * np.array([1, 1, 0.2])
    n = len(p)
    k = min(k, n - 1)
    order = rng.choice(["random", "rows", "morton", "as generated"])
    if order == "random":
        p = p[rng.permutation(n)]
    elif order == "rows":
        p = p[np.lexsort((p[:, 0], np.floor(p[:, 1] / (side / 40))))]
    elif order == "morton":
        q = np.floor((p[:, :2] - p[:, :2].min(0)) / (np.ptp(p[:, :2], axis=0) + 1e-9) * 1023).astype(np.int64)
        key = np.zeros(n, np.int64)
        for b in range(10):
            key |= ((q[:, 0] >> b) & 1) << (2 * b) | ((q[:, 1] >> b) & 1) << (2 * b + 1)
        p = p[np.argsort(key, kind="stable")]
    if rng.random() < 0.3:
        p = p + np.array([2647.0, 1177.0, 1500.0])
    xyz = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda()
    spacing = side / np.sqrt(n)
    res = float(spacing * rng.choice([1.5, 3.0, 8.0, 17.0, 40.0]))
    os.environ.pop("F4L_SV_EXACT_HOST", None)
    t0 = time.perf_counter()
    try:
        lab_d, K_d = engine.supervoxel(xyz, k, res)
    except F4LError as e:  # (a neighbour graph with more components than the target count: the reference's loop never returns on it)
        print(f"case {seed0 + case:4d} {kind:10s} {order:12s} n={n:7d} k={k:2d} res={res:8.3f}  REFUSED ({str(e)[-48:]})", flush=True)
        refused = globals().get("refused", 0) + 1
        continue
    torch.cuda.synchronize(); t_d = time.perf_counter() - t0
    os.environ["F4L_SV_EXACT_HOST"] = "1"
    lab_h, K_h = engine.supervoxel(xyz, k, res)
    os.environ.pop("F4L_SV_EXACT_HOST")
    diff = int((lab_d != lab_h).sum())
    ok = diff == 0 and K_d == K_h
    bad += not ok
    print(f"case {seed0 + case:4d} {kind:10s} {order:12s} n={n:7d} k={k:2d} res={res:8.3f} K={K_d:7d} {1e3 * t_d:8.1f} ms  "
          f"{'ok' if ok else f'MISMATCH: {diff} labels differ, K {K_d} / {K_h}'}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES", f"({globals().get('refused', 0)} clouds refused)")
sys.exit(1 if bad else 0)
