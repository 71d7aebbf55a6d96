"""Randomised check of the kNN lane kernel (four groups of 16 lanes with candidate blocks of their own) against the wave-per-query
search of round 1 (F4L_KNN_WAVE_PER_QUERY: identical indices, bit-equal d2, bit-equal fused normals) and against scipy's KD-tree
(distances), on clouds of random size and shape: surfaces, volumes, lines, lattices, clusters with duplicates, georeferenced
offsets, k = 1 .. 36, also n barely above k.   python3 tools/gpu/fuzz_knn.py [cases] [seed] [big]"""
import os, sys
import numpy as np, torch
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    k = int(rng.integers(1, 37))
    n = int(rng.choice([k + 1, k + 5, 63, 64, 65, 200, 1000, 5000, 40_000, 150_000] if len(sys.argv) <= 3 else [600_000, 2_000_000]))  # (third argument: large clouds)
    n = max(n, k + 1)
    kind = rng.choice(["surface", "volume", "line", "lattice", "clusters", "strip"])
    if kind == "surface":
        xy = rng.uniform(0, 50, (n, 2)); p = np.c_[xy, np.sin(xy[:, 0] / 7) * np.cos(xy[:, 1] / 5) + rng.normal(0, 0.01, n)]
    elif kind == "volume":
        p = rng.uniform(0, 10, (n, 3))
    elif kind == "line":
        t = rng.uniform(0, 100, n); p = np.c_[t, 0.3 * t + rng.normal(0, 1e-3, n), rng.normal(0, 1e-3, n)]
    elif kind == "lattice":
        m = int(np.ceil(n ** 0.5)); g = np.stack(np.meshgrid(np.arange(m), np.arange(m)), -1).reshape(-1, 2)[:n]; p = np.c_[g * 0.25, np.zeros(n)]
    elif kind == "clusters":
        c = rng.uniform(0, 30, (max(n // 50, 1), 3)); p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.05, (n, 3)); p[: n // 10] = p[0]  # duplicates
    else:
        p = np.c_[rng.uniform(0, 400, n), rng.uniform(0, 0.5, n), rng.normal(0, 0.01, n)]
    if rng.random() < 0.3:
        p = p + np.array([2647000.0, 1177000.0, 1500.0]) * (0.001 if rng.random() < 0.5 else 1.0)
    x = torch.from_numpy(p.astype(np.float32)).cuda()
    with_normals = k >= 3 and rng.random() < 0.5
    def run():
        if with_normals:
            i, nr, d = engine.knn_normals(x, k, return_d2=True); return i, d, nr
        i, d = engine.knn(x, k, return_d2=True); return i, d, None
    i1, d1, n1 = run()
    os.environ["F4L_KNN_WAVE_PER_QUERY"] = "1"
    i2, d2, n2 = run()
    del os.environ["F4L_KNN_WAVE_PER_QUERY"]
    flags = {"d2": torch.equal(d1, d2)}
    # indices may differ only inside groups of exactly equal distances -- both paths order those by index, so they must not
    flags["idx"] = torch.equal(i1, i2)
    if with_normals:
        # (k coincident neighbours have no covariance: the reference's formula gives NaN there, on both paths)
        flags["normals"] = torch.equal(torch.nan_to_num(n1, nan=7.0), torch.nan_to_num(n2, nan=7.0)) and torch.equal(torch.isnan(n1), torch.isnan(n2))
    xs = x.cpu().numpy().astype(np.float64)
    if n <= 700_000:
        dd, _ = cKDTree(xs).query(xs, k=k)
        dd = dd.reshape(n, k)
        flags["kdtree"] = bool(np.allclose(np.sqrt(d1.cpu().numpy()), dd, rtol=1e-12, atol=1e-12))
    # the row starts with the point itself, or -- coincident points -- with a lower index at distance 0
    first = i1[:, 0].cpu().numpy()
    flags["self"] = bool(((first == np.arange(n)) | ((first < np.arange(n)) & (d1[:, 0].cpu().numpy() == 0.0))).all())
    ok = all(flags.values())
    bad += not ok
    print(f"case {seed0 + case:4d} {kind:8s} n={n:7d} k={k:2d} normals={int(with_normals)}  {'ok' if ok else 'MISMATCH ' + str([f for f, v in flags.items() if not v])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
