"""CPU suite: the pointer-octree restatement of the reference's `Piecewise_ICP` (oracle/piecewise_octree.py; a16's checker).
Open3D is not installable here, so this oracle is PARITY UNPINNED; these tests hold it to the definitions it restates and to
hand-computable cases."""
import numpy as np

from oracle import piecewise_octree as PO


def test_octree_cells_and_traversal_order():
    # 8 clusters, one per octant of the unit cube, 300 points each: depth 1 -> 8 leaves visited in x + 2 y + 4 z order
    rng = np.random.default_rng(0)
    pts, want = [], []
    for k in range(8):
        lo = np.array([(k >> a) & 1 for a in range(3)], dtype=np.float64) * 0.5
        p = lo + rng.uniform(0.05, 0.45, (300, 3))
        pts.append(p)
    pts.append(np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]))  # bounding box corners; the max corner is out of bound
    pts = np.concatenate(pts)
    tree = PO.Octree(pts, 1)
    assert np.allclose(tree.origin, 0.0) and np.isclose(tree.size, 1.0)
    seen = []
    tree.traverse(lambda node: seen.append((node.leaf, len(node.indices))) and False)
    assert seen[0] == (False, 2401)  # the root saw every point but (1, 1, 1)
    assert [n for leaf, n in seen[1:]] == [301] + [300] * 7 and all(leaf for leaf, _ in seen[1:])
    for k in range(8):
        leaf = tree.locate_leaf(pts[300 * k + 5])
        assert set(range(300 * k, 300 * k + 300)) <= set(leaf.indices)
    assert tree.locate_leaf([1.0, 1.0, 1.0]) is None and tree.locate_leaf([2.0, 0.1, 0.1]) is None
    # early stop: a callback returning True on the root hides every leaf
    seen = []
    tree.traverse(lambda node: seen.append(node.leaf) or True)
    assert seen == [False]


def test_piecewise_icp_known_answer():
    """A flat 8 x 8 m plate of 4 x 4 cells; the second epoch equals the first except one cell lifted by 0.5 m: that cell's
    centroid distance exceeds mean + std, its points come out shifted by exactly the centroid difference, all others
    unchanged; rows: stable cells in lexicographic order of their centroids, then the unstable one."""
    rng = np.random.default_rng(1)
    src = np.c_[rng.uniform(0, 8, (40_000, 2)), rng.normal(0, 0.001, 40_000)]
    tgt = src.copy()
    lifted = (src[:, 0] >= 4) & (src[:, 0] < 6) & (src[:, 1] >= 2) & (src[:, 1] < 4)
    tgt[lifted, 2] += 0.5
    out = PO.piecewise_icp(src, tgt, smax=2.0, number_points_min=50)
    assert out["depth"] == 2 and out["n_centroids"] == 16 and out["n_stable_centroids"] == 15
    dv = out["dvfs"]
    moved = np.abs(dv[:, 5] - dv[:, 2]) > 1e-12
    assert moved.sum() == lifted.sum() and np.allclose(dv[moved, 5] - dv[moved, 2], 0.5, atol=1e-9)
    assert np.array_equal(dv[~moved, :3], dv[~moved, 3:]) and moved[-lifted.sum():].all()  # the unstable cell comes last
    assert out["n_stable_points"] == (~moved).sum()
    # stable cells in np.unique(axis=0) order of their centroids (src/piecewise_icp.py:166)
    lo, hi = np.minimum(src.min(0), tgt.min(0)), np.maximum(src.max(0), tgt.max(0))
    half = float(((hi - lo) / 2).max())
    origin, edge = np.minimum(lo, (lo + hi) / 2 - half), 2 * half / 4  # the octree's cube and its depth-2 cell edge
    ij = np.floor((dv[~moved, :2] - origin[:2]) / edge).astype(int)
    cell = ij[:, 0] * 4 + ij[:, 1]
    first = cell[np.sort(np.unique(cell, return_index=True)[1])]  # cells in order of appearance
    cen = np.array([dv[~moved][cell == c, :3].mean(axis=0) for c in first])
    assert np.array_equal(np.unique(cen, axis=0), cen)
    assert out["visualize"][0, 3] == 0 and out["visualize"][1, 3] == 10 and np.allclose(out["dvfms"][:, 3], np.abs(dv[:, 5] - dv[:, 2]))
