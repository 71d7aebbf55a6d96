"""Randomised check of the batched loop body (src/fine_matching.py::fine_matching_3d) against the patch-by-patch replay of the
reference's loop that tests/test_gpu_fine_matching.py holds (`_replay`: src/coarse_to_fine_matching_base.py:3254-3436 written with
the oracle's per-patch functions): random scenes and random settings -- matching mode (3D / 2D / fusion), `weighting_svd`, the
quality check and its thresholds, `num_min_fine_match`, the ICP threshold, both assign types, tgt2src rows.
    python3 tools/gpu/fuzz_fine_matching.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_fine_matching import _matches_from_2d, _replay, _scene, dev  # noqa: E402
from fusion4landslide_amd.src.fine_matching import fine_matching_3d  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    cells = int(rng.choice([3, 6, 9]))
    n = int(cells * cells * rng.choice([60, 250, 500]))
    src, tgt, so, soff, to, toff, corr = _scene(seed=int(rng.integers(0, 1000)), n=n, cells=cells)
    matching = str(rng.choice(["only_3d", "only_2d", "fusion"]))
    weighting = bool(matching == "fusion" and rng.random() < 0.5)
    corr2d = _matches_from_2d(src, tgt, seed=int(rng.integers(0, 100))) if matching != "only_3d" else None
    low_quality = bool(rng.random() < 0.5)
    n_check = int(rng.choice([10, 60]))
    tdd, tir = float(rng.choice([0.02, 0.03, 0.05])), float(rng.choice([0.3, 0.5, 0.7]))
    kw = dict(num_min_fine_match=int(rng.choice([3, 10, 30])), icp_threshold=float(rng.choice([0.05, 0.1, 0.2])),
              assign_type=str(rng.choice(["assign_all_src", "assign_then_nn"])), output_tgt2src=bool(rng.random() < 0.5), matching=matching,
              weighting_svd=weighting)
    res = fine_matching_3d(dev(src), dev(tgt), dev(so), dev(soff), dev(to), dev(toff), dev(corr), corr_tgt_2d=None if corr2d is None else dev(corr2d),
                           remove_low_quality_patch_matches=low_quality, num_min_matches_for_quality_check=n_check, thres_dist_diff=tdd,
                           thres_inlier_ratio=tir, median_max_resolution=0.03, **kw)
    dense, sparse, t2s, useful, glob, metric, Ts = _replay(src, tgt, so, soff, to, toff, corr, remove_low_quality=low_quality, n_check=n_check,
                                                           thres_dist_diff=tdd, thres_inlier_ratio=tir, median_res=0.03, corr2d=corr2d, **kw)
    P = len(soff) - 1
    flags = {"masks": bool(np.array_equal(res["mask_useful"].cpu().numpy(), useful) and np.array_equal(res["mask_global"].cpu().numpy(), glob))}
    it = res["iters"].cpu().numpy()
    flags["registered"] = set(np.nonzero(it >= 0)[0]) == set(Ts)
    if low_quality and flags["masks"]:
        flags["metric"] = bool(np.allclose(res["metric"].cpu().numpy(), metric.reshape(P, 2), rtol=1e-9, atol=1e-12))
    Tg = res["T"].cpu().numpy()
    worst, stepped, posed = 0.0, 0, 0
    for i, (T, fit, rmse) in Ts.items():
        s = src[so[soff[i]:soff[i + 1]]].astype(np.float64)
        e = float(np.abs(s @ T[:3, :3].T + T[:3, 3] - (s @ Tg[i, :3, :3].T + Tg[i, :3, 3])).max()) if len(s) else 0.0
        # a match of a handful of pairs leaves ICP under-determined: the two sides may settle apart (tools/gpu/fuzz_icp.py)
        if int(res["n_pairs"][i].sum()) < 40 or fit < 0.5:
            continue
        posed += 1
        worst = max(worst, e)
        stepped += e > 1e-9
    flags["transforms"] = worst <= 1e-6 and stepped <= max(1, posed // 25)

    def close(got, want):
        got = got.cpu().numpy()
        if got.shape != want.shape:
            return False
        tol = 2e-6 * np.abs(want).max() + 1e-6 if want.size else 0.0
        return bool(np.abs(got.astype(np.float64) - want.astype(np.float64)).max(initial=0.0) <= tol)

    if posed == len(Ts) and flags["transforms"] and stepped == 0:  # (rows are compared where every transform is pinned)
        flags["dense"] = close(res["dense"], dense)
        if kw["assign_type"] == "assign_all_src":
            flags["sparse"] = close(res["sparse"], sparse)
        if kw["output_tgt2src"]:
            flags["tgt2src"] = close(res["tgt2src"], t2s)
    else:
        flags["dense shape"] = tuple(res["dense"].shape) == dense.shape
    ok = all(flags.values())
    bad += not ok
    print(f"case {seed0 + case:4d} P={P:3d} n={n:6d} {matching:8s} w={int(weighting)} check={int(low_quality)} {kw['assign_type']:15s} min={kw['num_min_fine_match']:2d} "
          f"r={kw['icp_threshold']:.2f} registered {len(Ts):3d} (pinned {posed:3d}, worst {worst:.1e})  {'ok' if ok else 'MISMATCH ' + str([f for f, v in flags.items() if not v])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
