"""Randomised check of the device segmentation (f4l_supervoxel_parallel: kNN, normals and the segmentation on the GPU) against its
numpy restatement (oracle/sv_parallel.py) on clouds of random size, shape, density, resolution and k: identical labels,
representatives, counts, starting lambda, rounds and sweeps; K = occupied cells; labels contiguous, non-empty; the exchange's
fixed point.   python3 tools/gpu/fuzz_supervoxel.py [cases] [seed] [big]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine
from oracle import sv_parallel as M

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([400, 3000, 12_000, 40_000] if len(sys.argv) <= 3 else [150_000, 400_000]))  # (third argument: large clouds)
    k = int(rng.choice([8, 16, 30]))
    kind = rng.choice(["surface", "rough", "volume", "two sheets", "strip"])
    side = float(rng.choice([3.0, 10.0, 40.0]))
    if kind == "surface":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, 0.1 * side * np.sin(xy[:, 0] * 6 / side) * np.cos(xy[:, 1] * 4 / side)]
    elif kind == "rough":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, rng.normal(0, 0.02 * side, n)]
    elif kind == "volume":
        p = rng.uniform(0, side, (n, 3)) * np.array([1, 1, 0.3])
    elif kind == "two sheets":
        xy = rng.uniform(0, side, (n, 2)); p = np.c_[xy, np.where(rng.random(n) < 0.5, 0.0, 0.15 * side) + rng.normal(0, 0.002 * side, n)]
    else:
        p = np.c_[rng.uniform(0, 8 * side, n), rng.uniform(0, 0.1 * side, n), rng.normal(0, 0.003 * side, n)]
    if rng.random() < 0.3:
        p = p + np.array([2647.0, 1177.0, 1500.0])
    xyz = p.astype(np.float32)
    spacing = side / np.sqrt(n)
    res = float(spacing * rng.choice([3.0, 8.0, 17.0, 40.0]))
    t0 = time.perf_counter()
    try:
        labels, K, knn, nrm, reps, info = engine.supervoxel_parallel(torch.from_numpy(xyz).cuda(), k, res, return_intermediates=True)
    except RuntimeError as e:  # (status bits 2 / 4: the call says so instead of returning a partition that is not final)
        idx, nr = engine.knn_normals(torch.from_numpy(xyz).cuda(), k)
        ref = M.segment(xyz, nr.cpu().numpy(), idx.cpu().numpy(), res)
        print(f"case {seed0 + case:4d} {kind:10s} n={n:6d} k={k:2d} res={res:7.3f}  REFUSED ({str(e)[-60:]}); the model needs {ref['sweeps']} sweeps, {ref['rounds']} rounds", flush=True)
        refused = globals().get("refused", 0) + 1
        continue
    ref = M.segment(xyz, nrm.cpu().numpy(), knn.cpu().numpy(), res)
    flags = {"K": K == ref["n_supervoxels"] == M.occupied_cells(xyz, res), "status": int(info[2]) == 0 or int(info[2]) == 1,
             "labels": bool(np.array_equal(labels.cpu().numpy(), ref["labels"])), "reps": bool(np.array_equal(reps.cpu().numpy(), ref["reps"])),
             "lambda0": engine.supervoxel_lambda0(info) == ref["lambda0"], "rounds": int(info[6]) == ref["rounds"], "sweeps": int(info[3]) == ref["sweeps"]}
    inv = M.check_invariants(xyz, nrm.cpu().numpy(), knn.cpu().numpy().astype(np.int64), res, labels.cpu().numpy(), reps.cpu().numpy())
    flags["invariants"] = bool(inv["labels_contiguous"] and inv["all_non_empty"] and inv["fixed_point_violations"] == 0)
    ok = all(flags.values())
    bad += not ok
    print(f"case {seed0 + case:4d} {kind:10s} n={n:6d} k={k:2d} res={res:7.3f} K={K:6d} status={int(info[2])} {time.perf_counter() - t0:5.1f} s  "
          f"{'ok' if ok else 'MISMATCH ' + str([f for f, v in flags.items() if not v])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES", f"({globals().get('refused', 0)} clouds refused)")
