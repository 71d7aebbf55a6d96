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
