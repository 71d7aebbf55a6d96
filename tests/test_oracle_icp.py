"""CPU checks of the ICP oracle (oracle/f4l_oracle.c: orc_icp / orc_piecewise_icp).

Open3D 0.19.0, whose registration_icp the reference calls (utils/o3d_tools.py:46-50), is not installable here, so the
oracle's ICP is "parity unpinned" (DESIGN.md section 4).  What pins it instead:
  (i)   planted-motion known answers,
  (ii)  the definitions of fitness and inlier_rmse checked from the returned correspondences,
  (iii) an independent float64 restatement in numpy + scipy.cKDTree that must walk the same trajectory,
  (iv)  fixtures from a real Open3D, when somebody has produced them with tools/dump_o3d_goldens.py.
"""
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import oracle as O
from _util import rot_from_axis_angle


def _surface(rng, n, side=2.0, noise=0.0):
    xy = rng.uniform(0, side, (n, 2))
    z = 0.25 * np.sin(2.1 * xy[:, 0]) * np.cos(1.7 * xy[:, 1]) + 0.05 * np.sin(9 * xy[:, 0] + 1) * np.sin(7 * xy[:, 1])
    return np.c_[xy, z + rng.normal(0, noise, n)]


def _np_icp(src, tgt, T0, r, max_iter, icp_type="point2point", tgt_normals=None, rel=1e-6):
    """registration_icp as SURVEY.md a13 describes it [3P-knowledge], float64, numpy SVD / solve, KD-tree search."""
    tree = cKDTree(tgt)
    T = T0.copy()

    def evaluate(p):
        d, j = tree.query(p, k=1, distance_upper_bound=r)
        ok = np.isfinite(d)
        return ok, j, (ok.sum() / len(p) if len(p) else 0.0), (np.sqrt((d[ok] ** 2).sum() / ok.sum()) if ok.any() else 0.0)

    p = src @ T[:3, :3].T + T[:3, 3]
    ok, j, fit, rmse = evaluate(p)
    iters = 0
    for _ in range(max_iter):
        if not ok.any():
            break
        a, b = p[ok], tgt[j[ok]]
        U4 = np.eye(4)
        if icp_type == "point2point":
            ca, cb = a.mean(0), b.mean(0)
            H = (b - cb).T @ (a - ca) / len(a)  # Sigma of Umeyama (target x source)
            U, S, Vt = np.linalg.svd(H)
            D = np.diag([1.0, 1.0, 1.0 if np.linalg.det(U) * np.linalg.det(Vt) > 0 else -1.0])
            R = U @ D @ Vt
            U4[:3, :3], U4[:3, 3] = R, cb - R @ ca
        else:
            n = tgt_normals[j[ok]]
            J = np.c_[np.cross(a, n), n]
            res = ((a - b) * n).sum(1)
            x = np.linalg.solve(J.T @ J, -J.T @ res)
            al, be, ga = x[:3]
            Rz = np.array([[np.cos(ga), -np.sin(ga), 0], [np.sin(ga), np.cos(ga), 0], [0, 0, 1]])
            Ry = np.array([[np.cos(be), 0, np.sin(be)], [0, 1, 0], [-np.sin(be), 0, np.cos(be)]])
            Rx = np.array([[1, 0, 0], [0, np.cos(al), -np.sin(al)], [0, np.sin(al), np.cos(al)]])
            U4[:3, :3], U4[:3, 3] = Rz @ Ry @ Rx, x[3:]
        T = U4 @ T
        p = p @ U4[:3, :3].T + U4[:3, 3]
        pf, pr = fit, rmse
        ok, j, fit, rmse = evaluate(p)
        iters += 1
        if abs(pf - fit) < rel and abs(pr - rmse) < rel:
            break
    return T, fit, rmse, iters, ok, j


@pytest.mark.parametrize("icp_type", ["point2point", "point2plane"])
def test_icp_oracle_walks_the_same_trajectory_as_numpy_restatement(icp_type):
    rng = np.random.default_rng(3)
    for case in range(4):
        tgt = _surface(rng, 900, noise=0.002)
        src0 = _surface(rng, 600)
        src0 = src0[(src0[:, 0] > 0.15) & (src0[:, 0] < 1.85) & (src0[:, 1] > 0.15) & (src0[:, 1] < 1.85)]
        R0 = rot_from_axis_angle(rng.normal(size=3), 0.004 * (case + 1))
        src = src0 @ R0.T + rng.uniform(-0.02, 0.02, 3)
        normals = O.o3d_estimate_normals(tgt, 30) if icp_type == "point2plane" else None
        T0 = np.eye(4)
        ref = _np_icp(src, tgt, T0, 0.1, 30, icp_type, normals)
        got = O.icp(src, tgt, T0, max_corr_dist=0.1, max_iter=30, icp_type=icp_type, tgt_normals=normals)
        assert got["iters"] == ref[3]
        assert np.abs(got["est_transform"] - ref[0]).max() <= 1e-9
        assert abs(got["fitness"] - ref[1]) == 0.0 and abs(got["inlier_rmse"] - ref[2]) <= 1e-12
        cs = got["correspondence_set"]
        assert np.array_equal(cs[:, 0], np.nonzero(ref[4])[0]) and np.array_equal(cs[:, 1], ref[5][ref[4]])


def test_icp_oracle_point2plane_far_from_the_origin_is_the_extended_precision_step():
    """orc_point2plane sums and solves the 6 x 6 equations in long double: 3 km from the origin one pass of the oracle is the
    numpy `longdouble` pass to 1e-10 m, where the same pass with double sums is 1e-8 m away (its own rounding: the entries are
    |origin|^2 per pair, the information is in their patch-sized variation).  And a step with fewer pairs than unknowns is not
    taken (the kernel's rule)."""
    rng = np.random.default_rng(8)
    origin = np.array([2647.0, 1177.0, 1500.0])
    tgt = (_surface(rng, 6000, noise=0.002) + origin).astype(np.float32).astype(np.float64)
    src0 = _surface(rng, 5000)
    src0 = src0[(src0[:, 0] > 0.15) & (src0[:, 0] < 1.85) & (src0[:, 1] > 0.15) & (src0[:, 1] < 1.85)]
    src = ((src0 @ rot_from_axis_angle(rng.normal(size=3), 0.006).T + rng.uniform(-0.02, 0.02, 3)) + origin).astype(np.float32).astype(np.float64)
    normals = O.o3d_estimate_normals(tgt, 30)
    got = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane", tgt_normals=normals, fixed_iters=True)
    d, j = cKDTree(tgt).query(src)
    ok = d < 0.1

    def one_pass(dtype):
        a, b, n = src[ok].astype(dtype), tgt[j[ok]].astype(dtype), normals[j[ok]].astype(dtype)
        J = np.concatenate([np.cross(a, n), n], axis=1)
        res = ((a - b) * n).sum(1)
        A, rhs = np.zeros((6, 6), dtype), np.zeros(6, dtype)
        for i in range(len(a)):  # (one pair after the other, like the oracle)
            A += np.outer(J[i], J[i])
            rhs -= J[i] * res[i]
        M = np.concatenate([A.astype(np.longdouble), rhs.astype(np.longdouble)[:, None]], axis=1)
        for c in range(6):
            pv = c + int(np.argmax(np.abs(M[c:, c])))
            M[[c, pv]] = M[[pv, c]]
            for r_ in range(c + 1, 6):
                M[r_] -= M[r_, c] / M[c, c] * M[c]
        x = np.zeros(6, np.longdouble)
        for r_ in range(5, -1, -1):
            x[r_] = (M[r_, 6] - M[r_, r_ + 1:6] @ x[r_ + 1:]) / M[r_, r_]
        x = x.astype(np.float64)
        al, be, ga = x[:3]
        Rz = np.array([[np.cos(ga), -np.sin(ga), 0], [np.sin(ga), np.cos(ga), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(be), 0, np.sin(be)], [0, 1, 0], [-np.sin(be), 0, np.cos(be)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(al), -np.sin(al)], [0, np.sin(al), np.cos(al)]])
        return src @ (Rz @ Ry @ Rx).T + x[3:]

    T = got["est_transform"]
    moved = src @ T[:3, :3].T + T[:3, 3]
    e_ld, e_d = np.abs(moved - one_pass(np.longdouble)).max(), np.abs(moved - one_pass(np.float64)).max()
    assert e_ld <= 1e-10 and e_d >= 5 * e_ld, (e_ld, e_d)
    few = O.icp(src[:5], tgt, np.eye(4), max_corr_dist=0.1, max_iter=30, icp_type="point2plane", tgt_normals=normals)
    assert np.array_equal(few["est_transform"], np.eye(4)) and few["fitness"] == 1.0 and few["iters"] == 1


def test_icp_oracle_recovers_planted_motion_and_reports_definitional_scores():
    rng = np.random.default_rng(11)
    tgt = _surface(rng, 4000)
    keep = (tgt[:, 0] > 0.2) & (tgt[:, 0] < 1.8) & (tgt[:, 1] > 0.2) & (tgt[:, 1] < 1.8)
    R = rot_from_axis_angle([0.3, -0.2, 1.0], 0.01)
    t = np.array([0.012, -0.018, 0.009])
    src = (tgt[keep] - t) @ R  # tgt = R src + t exactly, point for point
    out = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=30)
    T = out["est_transform"]
    assert np.abs(T[:3, :3] - R).max() <= 1e-5 and np.abs(T[:3, 3] - t).max() <= 1e-5
    cs = out["correspondence_set"]
    moved = src @ T[:3, :3].T + T[:3, 3]
    d = np.linalg.norm(moved[cs[:, 0]] - tgt[cs[:, 1]], axis=1)
    assert out["fitness"] == len(cs) / len(src) == 1.0
    assert abs(out["inlier_rmse"] - np.sqrt((d ** 2).mean())) <= 1e-12
    assert (d < 0.1).all()
    # every correspondence is the nearest neighbour of the moved point
    assert np.array_equal(cKDTree(tgt).query(moved)[1][cs[:, 0]], cs[:, 1])


def test_icp_oracle_edge_cases_and_batch_equals_single_calls():
    rng = np.random.default_rng(12)
    a, b = _surface(rng, 300), _surface(rng, 300)
    # nothing within range: the init comes back, zero scores, no iteration counted as converged work
    far = O.icp(a, b + 5.0, np.eye(4), max_corr_dist=0.1, max_iter=30)
    assert np.array_equal(far["est_transform"], np.eye(4)) and far["fitness"] == 0.0 and far["inlier_rmse"] == 0.0
    assert len(far["correspondence_set"]) == 0
    # batch over ragged patches == one call per patch (float32 storage promoted like Vector3dVector does)
    src = np.concatenate([a[:120], a[120:121], a[121:]]).astype(np.float32)
    tgt = np.concatenate([b[:200], b[200:]]).astype(np.float32)
    soff, toff = np.array([0, 120, 121, 300, 300], np.int64), np.array([0, 200, 200, 300, 300], np.int64)
    out = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    for p in range(4):
        s, t = src[soff[p]:soff[p + 1]].astype(np.float64), tgt[toff[p]:toff[p + 1]].astype(np.float64)
        if len(s) == 0 or len(t) == 0:
            assert np.array_equal(out["T"][p], np.eye(4)) and out["fitness"][p] == 0.0
            continue
        one = O.icp(s, t, np.eye(4), max_corr_dist=0.1, max_iter=30)
        assert np.array_equal(out["T"][p], one["est_transform"]) and out["iters"][p] == one["iters"]
    # threads do not change the answer
    O.set_threads(4)
    try:
        par = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    finally:
        O.set_threads(1)
    assert np.array_equal(par["T"], out["T"]) and np.array_equal(par["iters"], out["iters"])


def test_icp_oracle_vs_open3d_goldens_when_present(golden_dir):
    """tools/dump_o3d_goldens.py, run where Open3D 0.19.0 is installed, writes tests/golden/o3d_icp_golden.npz; with it
    the oracle is pinned against the reference's actual arithmetic.  Absent here: skipped (parity unpinned)."""
    path = os.path.join(golden_dir, "o3d_icp_golden.npz")
    if not os.path.exists(path):
        pytest.skip("no Open3D fixtures (tools/dump_o3d_goldens.py needs Open3D 0.19.0)")
    g = np.load(path)
    for c in range(int(g["n_cases"])):
        src, tgt = g[f"src_{c}"], g[f"tgt_{c}"]
        for icp_type in ("point2point", "point2plane"):
            out = O.icp(src, tgt, g[f"init_{c}"], max_corr_dist=float(g["threshold"]), max_iter=30, icp_type=icp_type)
            T = g[f"T_{icp_type}_{c}"]
            moved_a = src @ out["est_transform"][:3, :3].T + out["est_transform"][:3, 3]
            moved_b = src @ T[:3, :3].T + T[:3, 3]
            assert np.abs(moved_a - moved_b).max() <= 1e-6, (c, icp_type)  # reference prints %.6f
            assert abs(out["fitness"] - float(g[f"fitness_{icp_type}_{c}"])) <= 1e-3
            assert abs(out["inlier_rmse"] - float(g[f"rmse_{icp_type}_{c}"])) <= 1e-5
