"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def knn_equal_within_ties(idx_a, idx_b, d2_a, d2_b=None, rtol=0.0):
    """kNN lists agree when, per query, the d2 sequences agree and the index *sets* agree inside every
    group of equal d2 (kd_tree.h:90-111 orders exact ties by traversal).  A tie group cut by the k-th
    slot may legitimately hold different members, so that last group is only checked for its d2."""
    idx_a, idx_b = np.asarray(idx_a), np.asarray(idx_b)
    if d2_b is not None:
        np.testing.assert_allclose(d2_a, d2_b, rtol=rtol, atol=0.0)
    n, k = idx_a.shape
    same = idx_a == idx_b
    bad_rows = np.nonzero(~same.all(axis=1))[0]
    for i in bad_rows:
        d = d2_a[i]
        j = 0
        while j < k:
            e = j
            while e + 1 < k and d[e + 1] == d[j]:
                e += 1
            if e == k - 1 and e > j or (e == k - 1 and not same[i, j:e + 1].all()):
                # group touches the k-th slot: members beyond k are interchangeable
                if e == j and idx_a[i, j] != idx_b[i, j]:
                    # single-element last group differing -> must be a tie with the (k+1)-th; accept only if d2 equal
                    pass
                j = e + 1
                continue
            if set(idx_a[i, j:e + 1].tolist()) != set(idx_b[i, j:e + 1].tolist()):
                return False, int(i)
            j = e + 1
    return True, -1


def rot_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def rotation_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def same_partition(la, lb):
    """Two labelings describe the same partition (labels may be permuted)."""
    la, lb = np.asarray(la), np.asarray(lb)
    pairs = np.unique(np.stack([la, lb], axis=1), axis=0)
    return pairs.shape[0] == np.unique(la).shape[0] == np.unique(lb).shape[0]


def partition_quality(xyz, nrm, lab):
    """RMS distance of the points to their supervoxel's centroid, mean normal deviation inside a supervoxel (1 - |n . mean
    normal|), coefficient of variation of the sizes."""
    xyz, lab = xyz.astype(np.float64), lab.astype(np.int64)
    K = lab.max() + 1
    cnt = np.bincount(lab, minlength=K).astype(float)
    c = np.stack([np.bincount(lab, weights=xyz[:, d], minlength=K) / cnt for d in range(3)], 1)
    rms = float(np.sqrt((np.linalg.norm(xyz - c[lab], axis=1) ** 2).mean()))
    first = np.zeros(K, dtype=np.int64)
    first[lab[::-1]] = np.arange(len(lab))[::-1]
    s = np.sign(np.sum(nrm * nrm[first][lab], axis=1))
    s[s == 0] = 1
    mn = np.stack([np.bincount(lab, weights=(nrm * s[:, None])[:, d], minlength=K) for d in range(3)], 1)
    mn /= np.maximum(np.linalg.norm(mn, axis=1, keepdims=True), 1e-300)
    return rms, float((1 - np.abs(np.sum(nrm * mn[lab], axis=1))).mean()), float(cnt.std() / cnt.mean())


def piecewise_motion_scene(xyz, block=0.25, seed=11, tmax=0.004, noise=0.0005, extent=1.0):
    """The second epoch of a golden cloud under a PIECEWISE rigid motion (SURVEY.md 8d's field at the cloud's scale): square
    blocks of side `block`, each with its own rotation <= 0.5 deg about a random axis through the block centre and its own
    translation U(-tmax, tmax), plus N(0, noise).  Returns (tgt (n, 3) float32, truth (n, 3) float64: the planted displacement
    of every point).  A patch that straddles a block boundary cannot follow both blocks: how well a partition's per-patch
    rigid fits recover `truth` depends on where the partition puts its boundaries."""
    rng = np.random.default_rng(seed)
    nb = int(np.ceil(extent / block))
    bx = np.minimum((xyz[:, 0] / block).astype(int), nb - 1)
    by = np.minimum((xyz[:, 1] / block).astype(int), nb - 1)
    bid = by * nb + bx
    B = nb * nb
    R = np.stack([rot_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 0.5))) for _ in range(B)])
    t = rng.uniform(-tmax, tmax, (B, 3))
    c = np.c_[(bx + 0.5) * block, (by + 0.5) * block, np.zeros(len(xyz))]
    p = xyz.astype(np.float64)
    moved = np.einsum("nij,nj->ni", R[bid], p - c) + c + t[bid]
    return (moved + rng.normal(0, noise, p.shape)).astype(np.float32), moved - p


# ---- the LARGE reference-pinned partition fixture (tests/golden/sv_large_ref.npz, tools/make_golden_supervoxel.py --large) -------
LARGE_CASE = dict(seed=17, n=300_000, extent=5.0, k=30, resolution=0.12)


def large_surface_cloud(seed, n, extent):
    """The fixture's cloud, regenerated from its seed (the fixture stores no coordinates and no neighbour lists: only what the
    reference computed).  numpy's PCG64 streams of `uniform` / `normal` are stable across the versions in use; the fixture also
    keeps a checksum of the float32 bits, checked before anything else is."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, extent, (n, 2))
    z = 0.4 * np.sin(2.1 * xy[:, 0]) * np.cos(1.7 * xy[:, 1]) + 0.05 * np.sin(9.0 * xy[:, 0] + 1.0) * np.sin(7.0 * xy[:, 1])
    z = z + np.where((xy[:, 0] > 0.6 * extent) & (xy[:, 1] > 0.5 * extent), 0.15, 0.0)  # a step: the normal term decides along it
    return np.c_[xy, z + rng.normal(0, 0.003, n)].astype(np.float32)


def bits_checksum(a):
    """Order-dependent 64-bit checksum of an array's bytes (sum of the 32-bit words times their 1-based position, mod 2^64)."""
    w = np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint32).astype(np.uint64)
    return int((w * (np.arange(1, len(w) + 1, dtype=np.uint64) | np.uint64(1))).sum(dtype=np.uint64))
