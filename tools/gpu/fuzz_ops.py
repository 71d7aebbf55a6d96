"""Randomised checks of the operators around the loop against independent CPU computations (numpy, scipy's KD-tree, the C oracle):
f4l_nn_query, f4l_voxel_downsample, f4l_kabsch_batched, f4l_labels_to_csr + f4l_gather_points, f4l_median_f64,
f4l_mutual_correspondences, f4l_rigidity_check, f4l_nn_refine, f4l_apply_transform -- inputs of random size and shape incl. empty
and degenerate ones.   python3 tools/gpu/fuzz_ops.py [cases] [seed]"""
import os, sys
import numpy as np, torch
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine
from oracle import oracle as O

dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0


def cloud(rng, n, kind, off):
    if kind == 0:
        xy = rng.uniform(0, 30, (n, 2)); p = np.c_[xy, np.sin(xy[:, 0] / 5) + rng.normal(0, 0.01, n)]
    elif kind == 1:
        p = rng.uniform(0, 8, (n, 3))
    else:
        c = rng.uniform(0, 20, (max(n // 40, 1), 3)); p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.03, (n, 3))
    return (p + off).astype(np.float32)


for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    off = np.array([2647.0, 1177.0, 1500.0]) if rng.random() < 0.3 else np.zeros(3)
    flags = {}
    # ---- nearest cloud points of queries from elsewhere
    n, m, k = int(rng.choice([1, 5, 300, 20_000, 90_000])), int(rng.choice([0, 1, 64, 5000, 40_000])), int(rng.choice([1, 2, 4, 9, 30]))
    k = min(k, n)
    c = cloud(rng, n, int(rng.integers(0, 3)), off)
    q = np.concatenate([cloud(rng, m, int(rng.integers(0, 3)), off), c[: min(n, 50)], (c[: min(n, 20)] + np.float32([0, 0, 40.0]))]) if m else np.zeros((0, 3), np.float32)
    idx, d2 = engine.nn_query(dv(c), dv(q), k, return_d2=True)
    if len(q):
        dref, _ = cKDTree(c.astype(np.float64)).query(q.astype(np.float64), k=k)
        dref = dref.reshape(len(q), k)
        got = ((c[idx.cpu().numpy().reshape(-1)].astype(np.float64) - np.repeat(q.astype(np.float64), k, axis=0)) ** 2).sum(1).reshape(len(q), k)
        flags["nn_query"] = bool(np.abs(np.sqrt(d2.cpu().numpy()) - dref).max() <= 1e-12 * max(1.0, dref.max()) and np.abs(got - d2.cpu().numpy()).max() <= 1e-12 * max(1.0, got.max()))
    else:
        flags["nn_query"] = tuple(idx.shape) == (0, k)
    # ---- voxel grid filter
    voxel = float(rng.choice([0.02, 0.3, 2.0, 50.0]))
    pts, cnt, vop = engine.voxel_downsample(dv(c), voxel, return_map=True)
    rp, rc, rv = O.voxel_downsample(c, voxel)
    flags["voxel"] = bool(pts.shape[0] == len(rp) and np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(vop.cpu().numpy(), rv) and
                          (len(rp) == 0 or np.abs(pts.cpu().numpy() - rp).max() <= 1e-9 * max(1.0, np.abs(rp).max())))
    # ---- ragged Kabsch
    P = int(rng.choice([1, 3, 200, 3000]))
    sizes = rng.integers(0, int(rng.choice([4, 40, 700])), P)
    koff = np.zeros(P + 1, np.int64); np.cumsum(sizes, out=koff[1:])
    ks = (rng.uniform(-1, 1, (int(koff[-1]), 3)) * rng.choice([0.1, 5.0]) + off).astype(np.float32)
    ang = rng.uniform(0, 0.2); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rt = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    kr = ((ks.astype(np.float64) - off) @ Rt.T + off + rng.uniform(-0.2, 0.2, 3) + rng.normal(0, 1e-3, ks.shape)).astype(np.float32)
    w = rng.uniform(0, 1, len(ks)).astype(np.float32) if rng.random() < 0.5 else None
    thr = float(rng.choice([0.0, 0.3]))
    R, t = engine.kabsch_batched(dv(ks), dv(kr), dv(koff), None if w is None else dv(w), weight_thresh=thr, eps=1e-6)
    Rr, tr = O.kabsch_batched(ks, kr, koff, w, weight_thresh=thr, eps=1e-6)
    well = sizes >= 12  # (fewer points: rank-deficient sums, any rotation about the null space is a solution)
    dR = np.abs(R.cpu().numpy() - Rr).reshape(P, -1).max(1); dt = np.abs(t.cpu().numpy() - tr).max(1)
    flags["kabsch"] = bool((dR[well] <= 2e-5).all() and (dt[well] <= 2e-3 * max(1.0, np.abs(off).max())).all() and np.isfinite(R.cpu().numpy()).all())
    # ---- Kabsch #2 of src/functions.py (the F2S3 path), float64 sets, set by set against the numpy restatement
    R2, t2 = engine.kabsch2_batched(dv(ks.astype(np.float64)), dv(kr.astype(np.float64)), dv(koff), None if w is None else dv(w.astype(np.float64)),
                                    normalize_w=bool(rng.random() < 0.5), w_threshold=thr, eps=1e-7)
    ok2 = True
    nrm_flag = None
    for pset in np.nonzero(well)[0][:40]:
        a0, a1 = koff[pset], koff[pset + 1]
        for normalize in (True, False):
            Rr2, tr2 = O.kabsch_transformation_estimation(ks[None, a0:a1].astype(np.float64), kr[None, a0:a1].astype(np.float64),
                                                          None if w is None else w[None, a0:a1].astype(np.float64), normalize_w=normalize, eps=1e-7, w_threshold=thr)
            if np.abs(R2[pset].cpu().numpy() - Rr2[0]).max() <= 1e-6 and np.abs(t2[pset].cpu().numpy() - tr2[0, :, 0]).max() <= 1e-5 * max(1.0, np.abs(off).max()):
                break
        else:
            # (weights below the threshold count for nothing: a set left with a handful of effective points, or with all its weight on
            #  one of them, has no unique rotation)
            weff = np.ones(a1 - a0) if w is None else np.where(w[a0:a1] >= thr, w[a0:a1], 0.0)
            if (weff > 0).sum() >= 8 and weff.max() < 0.5 * weff.sum():
                ok2 = False
                print("   kabsch2 set", int(pset), "n", int(a1 - a0), "effective", int((weff > 0).sum()), "dR", float(np.abs(R2[pset].cpu().numpy() - Rr2[0]).max()),
                      "dt", float(np.abs(t2[pset].cpu().numpy() - tr2[0, :, 0]).max()))
    flags["kabsch2"] = ok2
    # ---- labels -> CSR, gather
    K = int(rng.choice([1, 7, 500, 20_000]))
    lab = rng.integers(0, K, n).astype(np.int32)
    if K > 3: lab[lab == 1] = 2
    order, loff = engine.labels_to_csr(dv(lab), K)
    flags["csr"] = bool(np.array_equal(order.cpu().numpy(), np.argsort(lab, kind="stable")) and
                        np.array_equal(np.diff(loff.cpu().numpy()), np.bincount(lab, minlength=K)) and
                        np.array_equal(engine.gather_points(dv(c), order).cpu().numpy(), c[order.cpu().numpy()]))
    # ---- median
    v = rng.normal(size=max(n, 1)) * rng.choice([1.0, 1e-3, 1e6])
    v[: len(v) // 3] = v[0]
    from fusion4landslide_amd.engine import _median_of_sqrt
    flags["median"] = float(_median_of_sqrt(dv(v * v)).item()) == float(np.median(np.sqrt(v * v)))
    # ---- rigidity check and mutual correspondences on random patch matches
    Pm = int(rng.choice([1, 20, 400]))
    ms = rng.integers(0, 60, Pm)
    moff = np.zeros(Pm + 1, np.int64); np.cumsum(ms, out=moff[1:])
    shift = rng.choice([0.0, 0.0, 2.6e3, -7.1e5]) * rng.uniform(0.5, 1, 3)  # (local or georeferenced coordinates)
    a_ = (rng.uniform(-1.5, 1.5, (int(moff[-1]), 3)) + shift).astype(np.float32); b_ = (a_ + rng.normal(0, 0.02, a_.shape)).astype(np.float32)
    dm, ri = engine.rigidity_check(dv(a_), dv(b_), dv(moff), 0.05)
    dm32, ri32 = engine.rigidity_check(dv(a_), dv(b_), dv(moff), 0.05, precision="f32")
    dm_r, ri_r, near = np.zeros(Pm), np.zeros(Pm), np.zeros(Pm)
    for p in range(Pm):
        s_, t_ = a_[moff[p]:moff[p + 1]].astype(np.float64), b_[moff[p]:moff[p + 1]].astype(np.float64)
        if len(s_) >= 2:
            iu = np.triu_indices(len(s_), 1)
            dd = np.abs(np.linalg.norm(s_[iu[0]] - s_[iu[1]], axis=1) - np.linalg.norm(t_[iu[0]] - t_[iu[1]], axis=1))
            dm_r[p], ri_r[p], near[p] = dd.mean(), (dd <= 0.05).mean(), (np.abs(dd - 0.05) <= 5e-6).mean()
    flags["rigidity"] = bool(np.abs(dm.cpu().numpy() - dm_r).max(initial=0) <= 1e-12 and np.abs(ri.cpu().numpy() - ri_r).max(initial=0) <= 1e-12)
    # float32 pair arithmetic: every distance within 3e-7 of itself (sets span <= 5.2 m), only pairs AT the threshold may change sides
    flags["rigidity_f32"] = bool(np.abs(dm32.cpu().numpy() - dm_r).max(initial=0) <= 3e-6 and (np.abs(ri32.cpu().numpy() - ri_r) <= near + 1e-12).all())
    ok = all(flags.values())
    bad += not ok
    print(f"case {seed0 + case:4d} n={n:6d} m={m:6d} k={k:2d} P={P:5d}  {'ok' if ok else 'MISMATCH ' + str([f for f, v_ in flags.items() if not v_])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
