"""TEST INFRASTRUCTURE ONLY -- numpy model of the PARALLEL supervoxel segmentation of
fusion4landslide_amd/csrc/supervoxel_gpu.hip (f4l_supervoxel_segment_device).

The sequential algorithm of the reference (codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-248) is restated in
oracle/f4l_oracle.c and pinned by the reference's own code.  The parallel variant keeps its structure and criteria but fuses
the representatives of a round in conflict-free sub-rounds over the graph of representatives as the round's start found it (the
adjacency of absorbed representatives reaches their new representative with the next round's list), so its labels differ from
the reference's; it is deterministic,
and this independent restatement (vectorised numpy, written from the algorithm's description, not from the kernel code)
must reproduce the device's labels EXACTLY.  What ties the variant to the reference are the invariants checked in
tests/: K = occupied cells of the resolution grid, labels 0..K-1 all non-empty, the exchange's fixed point, and the
agreement of the downstream displacements.

Only tests/ may import this module.
"""
import numpy as np

LAMBDA_ROUNDS, SUBROUNDS, SWEEPS = 56, 3, 1024


def metric(xyz, nrm, a, b, resolution):
    """supervoxel.cpp:27-40, the operation order of sv_metric.h (double, no contraction)."""
    pa, pb = xyz[a].astype(np.float64), xyz[b].astype(np.float64)
    na, nb = nrm[a], nrm[b]
    dot = na[..., 0] * nb[..., 0] + na[..., 1] * nb[..., 1] + na[..., 2] * nb[..., 2]
    t1, t2, t3 = pa[..., 0] - pb[..., 0], pa[..., 1] - pb[..., 1], pa[..., 2] - pb[..., 2]
    return 1.0 - np.abs(dot) + np.sqrt(t1 * t1 + t2 * t2 + t3 * t3) / resolution * 0.4


def occupied_cells(xyz, resolution, grid_bbox=None):
    """grid_sample.h:48-68: number of occupied cells of the grid anchored at the bounding box minimum (of the cloud, or the
    box given: a slab counts its cells in the whole cloud's grid)."""
    p = xyz.astype(np.float64)
    if grid_bbox is None:
        mn, mx = p.min(axis=0), p.max(axis=0)
    else:
        mn, mx = (np.asarray(grid_bbox, dtype=np.float32).astype(np.float64)[i:i + 3] for i in (0, 3))
    size = ((mx - mn) / resolution + 1).astype(np.int64)
    c = np.clip(((p - mn) / resolution).astype(np.int64), 0, size - 1)
    key = (c[:, 0] * size[1] + c[:, 1]) * size[2] + c[:, 2]
    return int(np.unique(key).shape[0])


def _heads(v, rnd):
    with np.errstate(over="ignore"):
        h = (v.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ np.uint32(((rnd + 1) * 0x85EBCA6B) & 0xFFFFFFFF)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return (h & np.uint32(1)) != 0


def _f2ord(f32):
    u = f32.view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000))


def _find(parent):
    """Every point's root."""
    r = parent.copy()
    while True:
        rr = r[r]
        if np.array_equal(rr, r):
            return r
        r = rr


def segment(xyz, nrm, knn, resolution, grid_bbox=None):
    """Returns dict(labels (n,) int32, reps (K,) int32, n_supervoxels, K_target, status, sweeps, lambda0).  A knn entry < 0 (or
    equal to its row) is "no neighbour"."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    nrm = np.ascontiguousarray(nrm, dtype=np.float64)
    knn = np.ascontiguousarray(knn, dtype=np.int64)
    n, k = knn.shape
    knn = np.where(knn < 0, np.arange(n)[:, None], knn)
    K = occupied_cells(xyz, resolution, grid_bbox)
    idx = np.repeat(np.arange(n), k)
    flat = knn.reshape(-1)
    notself = flat != idx
    m_all = np.full(n * k, np.inf)
    m_all[notself] = metric(xyz, nrm, idx[notself], flat[notself], resolution)
    dis0 = m_all.reshape(n, k).min(axis=1)
    dis0[np.isinf(dis0)] = np.finfo(np.float64).max
    lam = max(np.finfo(np.float64).eps, float(np.sort(dis0)[n // 2]))
    lambda0 = lam

    parent = np.arange(n, dtype=np.int64)
    size = np.ones(n, dtype=np.int64)
    eu, ev = idx[notself], flat[notself]
    live, rnd, stalled, rounds = n, 0, False, 0
    for _ in range(LAMBDA_ROUNDS):
        if live <= K or stalled:
            break
        rounds += 1
        for _s in range(SUBROUNDS):
            if live <= K:
                break
            # (the list of a lambda round is the one its start built: an edge takes part while both its ends are still
            #  representatives; edges of absorbed ones wait for the next round's list)
            alive = (parent[eu] == eu) & (parent[ev] == ev)
            cu, cv = eu[alive], ev[alive]
            ok = _heads(cu, rnd) & ~_heads(cv, rnd)
            u, v = cu[ok], cv[ok]
            m = metric(xyz, nrm, u, v, resolution)
            el = (lam - size[v].astype(np.float64) * m) > 0.0
            u, v, m = u[el], v[el], m[el]
            rnd += 1
            if len(v) == 0:
                continue
            bestm = np.full(n, np.inf)
            np.minimum.at(bestm, v, m)
            tie = m == bestm[v]
            bestu = np.full(n, np.iinfo(np.int64).max)
            np.minimum.at(bestu, v[tie], u[tie])
            pv = np.nonzero(bestu != np.iinfo(np.int64).max)[0]
            pu = bestu[pv]
            loss = (size[pv].astype(np.float64) * bestm[pv]).astype(np.float32)
            key = (_f2ord(loss).astype(np.uint64) << np.uint64(32)) | pv.astype(np.uint64)
            budget = live - K
            if len(pv) > budget:
                sel = np.argsort(key, kind="stable")[:budget]
                pv, pu = pv[sel], pu[sel]
            parent[pv] = pu
            np.add.at(size, pu, size[pv])
            live -= len(pv)
        if live <= K:
            break
        parent = _find(parent)
        eu, ev = parent[eu], parent[ev]
        keep = eu != ev
        pairs = np.unique(np.stack([eu[keep], ev[keep]], axis=1), axis=0) if keep.any() else np.zeros((0, 2), np.int64)
        eu, ev = pairs[:, 0], pairs[:, 1]
        lam *= 2.0
        if len(eu) == 0:
            stalled = True
    parent = _find(parent)

    # boundary exchange: sweeps over the labels of the previous sweep
    lab = parent.copy()
    dis = metric(xyz, nrm, np.arange(n), lab, resolution)
    dirty = np.zeros(n, dtype=bool)
    full, on, sweeps = True, True, 0
    for _ in range(SWEEPS):
        if not on:
            break
        look = np.arange(n) if full else np.nonzero(dirty)[0]
        a = lab[look]
        best, bl = dis[look].copy(), a.copy()
        for j in range(k):
            b = lab[knn[look, j]]
            cand = (b != a) & (b != bl)
            if cand.any():
                d = np.full(len(look), np.inf)
                d[cand] = metric(xyz, nrm, look[cand], b[cand], resolution)
                better = cand & (d < best)
                best = np.where(better, d, best)
                bl = np.where(better, b, bl)
        ch = bl != a
        new_lab = lab.copy()
        new_lab[look[ch]] = bl[ch]
        dis[look[ch]] = best[ch]
        dirty = np.zeros(n, dtype=bool)
        dirty[look[ch]] = True
        dirty[knn[look[ch]].reshape(-1)] = True
        lab = new_lab
        sweeps += 1
        if ch.any():
            full = False
        elif not full:
            full = True
        else:
            on = False
    roots = np.nonzero(parent == np.arange(n))[0]
    rank = np.full(n, -1, dtype=np.int64)
    rank[roots] = np.arange(len(roots))
    status = (1 if stalled else 0) | (2 if (live > K and not stalled) else 0) | (4 if on else 0)
    return dict(labels=rank[lab].astype(np.int32), reps=roots.astype(np.int32), n_supervoxels=len(roots), K_target=K,
                status=status, sweeps=sweeps, lambda0=lambda0, rounds=rounds)


def check_invariants(xyz, nrm, knn, resolution, labels, reps):
    """The properties that tie a parallel partition to the reference's (hard parts).  Returns a dict of
    booleans / numbers; raises nothing."""
    xyz = np.asarray(xyz, dtype=np.float32)
    labels, reps = np.asarray(labels, dtype=np.int64), np.asarray(reps, dtype=np.int64)
    n, k = knn.shape
    K = occupied_cells(xyz, resolution)
    counts = np.bincount(labels, minlength=len(reps))
    dis = metric(xyz, nrm, np.arange(n), reps[labels], resolution)
    # fixed point of the exchange: no point has a neighbour whose representative is strictly closer
    worst = 0
    for j in range(k):
        b = labels[knn[:, j]]
        diff = b != labels
        d = np.full(n, np.inf)
        d[diff] = metric(xyz, nrm, np.nonzero(diff)[0], reps[b[diff]], resolution)
        worst += int((d < dis).sum())
    return dict(K_equals_cells=len(reps) == K, K=K, labels_contiguous=bool(labels.min() == 0 and labels.max() == len(reps) - 1),
                all_non_empty=bool((counts > 0).all()) and len(counts) == len(reps),
                reps_carry_own_label=bool((labels[reps] == np.arange(len(reps))).all()),
                reps_ascending=bool((np.diff(reps) > 0).all()), fixed_point_violations=worst,
                energy=float(dis.sum()))
