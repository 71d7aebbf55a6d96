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


def _relief(xy):
    return 0.25 * np.sin(2.1 * xy[:, 0]) * np.cos(1.7 * xy[:, 1]) + 0.05 * np.sin(9 * xy[:, 0] + 1) * np.sin(7 * xy[:, 1])


def _surface(rng, n, side=2.0, noise=0.0):
    xy = rng.uniform(0, side, (n, 2))
    return np.c_[xy, _relief(xy) + rng.normal(0, noise, n)]


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


def test_icp_oracle_robust_point2plane_far_from_the_origin_is_the_extended_precision_step():
    """The ROBUST variant (`point2plane_robust`, orc_point2plane_robust: this repository's rule, the kernel's default -- not
    Open3D's semantics, which are `point2plane`) sums and solves the 6 x 6 equations in long double: 3 km from the origin one
    pass of it is the numpy `longdouble` pass to 1e-10 m, where the same pass with double sums is 1e-8 m away (its own
    rounding: the entries are |origin|^2 per pair, the information is in their patch-sized variation).  And a step with fewer
    pairs than unknowns is not taken."""
    rng = np.random.default_rng(8)
    origin = np.array([2647.0, 1177.0, 1500.0])
    tgt = (_surface(rng, 6000, noise=0.002) + origin).astype(np.float32).astype(np.float64)
    src0 = _surface(rng, 5000)
    src0 = src0[(src0[:, 0] > 0.15) & (src0[:, 0] < 1.85) & (src0[:, 1] > 0.15) & (src0[:, 1] < 1.85)]
    src = ((src0 @ rot_from_axis_angle(rng.normal(size=3), 0.006).T + rng.uniform(-0.02, 0.02, 3)) + origin).astype(np.float32).astype(np.float64)
    normals = O.o3d_estimate_normals(tgt, 30)
    got = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane_robust", tgt_normals=normals, fixed_iters=True)
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
    few = O.icp(src[:5], tgt, np.eye(4), max_corr_dist=0.1, max_iter=30, icp_type="point2plane_robust", tgt_normals=normals)
    assert np.array_equal(few["est_transform"], np.eye(4)) and few["fitness"] == 1.0 and few["iters"] == 1
    # Open3D's own semantics take that step all the same (one correspondence is enough for it)
    few = O.icp(src[:5], tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane", tgt_normals=normals, fixed_iters=True)
    assert not np.array_equal(few["est_transform"], np.eye(4))
    # ... and in double: the strict restatement's single pass sits with the numpy float64 pass, not with the extended one
    strict = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane", tgt_normals=normals, fixed_iters=True)
    Ts = strict["est_transform"]
    moved_s = src @ Ts[:3, :3].T + Ts[:3, 3]
    assert np.abs(moved_s - one_pass(np.float64)).max() <= np.abs(moved_s - one_pass(np.longdouble)).max() + 1e-9


def _numpy_eigen_ldlt_solve(A, b):
    """Eigen::LDLT::compute + solve, statement by statement in numpy (independent of the C restatement): bordered
    factorisation, pivot = largest |stored diagonal| of the rows left (first on ties), pseudo-inverse of D."""
    A = np.array(A, dtype=np.float64)
    n = 6
    tr = np.arange(n)
    for k in range(n):
        big = k + int(np.argmax(np.abs(np.diag(A)[k:])))
        tr[k] = big
        if big != k:
            A[[k, big], :k] = A[[big, k], :k]
            A[big + 1:, [k, big]] = A[big + 1:, [big, k]]
            A[k, k], A[big, big] = A[big, big], A[k, k]
            for i in range(k + 1, big):
                A[i, k], A[big, i] = A[big, i], A[i, k]
        if k > 0:
            temp = np.diag(A)[:k] * A[k, :k]
            A[k, k] -= sum(A[k, j] * temp[j] for j in range(k))
            for i in range(k + 1, n):
                A[i, k] -= sum(A[i, j] * temp[j] for j in range(k))
        if abs(A[k, k]) > 0:
            A[k + 1:, k] /= A[k, k]
        elif k == 0:
            tr = np.arange(n)
            break
    y = np.array(b, dtype=np.float64)
    for k in range(n):
        y[[k, tr[k]]] = y[[tr[k], k]]
    for i in range(n):
        for j in range(i):
            y[i] -= A[i, j] * y[j]
    d = np.diag(A)
    y = np.where(np.abs(d) > np.finfo(np.float64).tiny, y / np.where(d == 0, 1, d), 0.0)
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            y[i] -= A[j, i] * y[j]
    for k in range(n - 1, -1, -1):
        y[[k, tr[k]]] = y[[tr[k], k]]
    return y


def test_eigen_ldlt_restatement_three_ways(tmp_path):
    """x = A.ldlt().solve(b), the solve inside Open3D's SolveLinearSystemPSD: the oracle's C restatement, an independent numpy
    one, and the PRODUCT's register version (fusion4landslide_amd/csrc/ldlt6.h, compiled here as host code) give the same bits
    on full-rank, rank-deficient and exactly-zero-structured systems; on full-rank ones it is the solution."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "ldlt6_host.cpp"
    src.write_text('#include "ldlt6.h"\n'
                   'extern "C" void prod_ldlt6_solve(const double *A36, const double *b6, double *x6) {\n'
                   '    double A[6][6], b[6], x[6];\n'
                   '    for (int i = 0; i < 6; ++i) { b[i] = b6[i]; for (int j = 0; j < 6; ++j) A[i][j] = A36[6 * i + j]; }\n'
                   '    f4l::ldlt6_solve_eigen(A, b, x);\n'
                   '    for (int i = 0; i < 6; ++i) x6[i] = x[i];\n}\n'
                   'extern "C" void prod_to_caller(double *M36, double *b6, double o0, double o1, double o2) {\n'
                   '    double M[6][6], b[6];\n'
                   '    for (int i = 0; i < 6; ++i) { b[i] = b6[i]; for (int j = 0; j < 6; ++j) M[i][j] = M36[6 * i + j]; }\n'
                   '    f4l::p2plane_system_to_caller_frame(M, b, o0, o1, o2);\n'
                   '    for (int i = 0; i < 6; ++i) { b6[i] = b[i]; for (int j = 0; j < 6; ++j) M36[6 * i + j] = M[i][j]; }\n}\n')
    so = tmp_path / "ldlt6_host.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                           os.path.join(root, "fusion4landslide_amd", "csrc"), str(src), "-o", str(so)])
    L = ctypes.CDLL(str(so))
    f = np.ctypeslib.ndpointer(np.float64, flags="C")
    L.prod_ldlt6_solve.argtypes = [f, f, f]
    L.prod_to_caller.argtypes = [f, f, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    rng = np.random.default_rng(5)
    for it in range(400):
        J = rng.normal(size=(rng.integers(1, 12), 6)) * rng.uniform(0.1, 10, size=6)
        if it % 5 == 0:
            J[:, rng.integers(0, 6)] = 0.0   # an exactly zero row and column
        if it % 7 == 0:
            J[:, 2:5] = 0.0                  # the plane with normals (0, 0, 1): rank 3 with exact zeros
        A, b = J.T @ J, rng.normal(size=6)
        x_c = O.ldlt6_solve(A, b)
        x_p = np.empty(6)
        L.prod_ldlt6_solve(np.ascontiguousarray(A.reshape(36)), b, x_p)
        assert np.array_equal(x_c, x_p), it
        if it < 120:
            assert np.array_equal(x_c, _numpy_eigen_ldlt_solve(A, b)), it
        if J.shape[0] >= 6 and it % 5 and it % 7:
            assert np.allclose(A @ x_c, b, rtol=1e-8, atol=1e-8)
        if it % 7 == 0:
            assert (x_c[2:5] == 0.0).all()   # the pseudo-inverse of D: no motion along what the data does not see
    # the system summed about a patch origin, moved to the caller's origin, is the system summed there
    for it in range(50):
        m = 20
        s, n, r = rng.normal(size=(m, 3)), rng.normal(size=(m, 3)), rng.normal(size=m)
        n /= np.linalg.norm(n, axis=1)[:, None]
        o = rng.normal(size=3) * 100
        Jp, Jc = np.c_[np.cross(s, n), n], np.c_[np.cross(s + o, n), n]
        M, b = np.ascontiguousarray((Jp.T @ Jp).reshape(36)), -(Jp.T @ r)
        L.prod_to_caller(M, b, *o)
        assert np.allclose(M.reshape(6, 6), Jc.T @ Jc, rtol=1e-9, atol=1e-6) and np.allclose(b, -(Jc.T @ r), rtol=1e-9, atol=1e-6)


def test_icp_oracle_point2plane_semantics_part_only_where_the_system_is_singular():
    """`point2plane` (Open3D's step, strict) and `point2plane_robust` walk the same trajectory on patches that pin their six
    unknowns (1e-9 m, same iteration count near the origin); they part on a four-pair patch (robust: no step; Open3D: whatever
    the singular system gives) and on an exactly planar target with normals (0, 0, 1) (robust: no step; Open3D: the tilt and the
    lift the data does see, nothing along the three directions it does not -- Eigen's pseudo-inverse of D)."""
    rng = np.random.default_rng(21)
    tgt = _surface(rng, 1200, noise=0.002)
    src0 = _surface(rng, 700)
    src0 = src0[(src0[:, 0] > 0.15) & (src0[:, 0] < 1.85) & (src0[:, 1] > 0.15) & (src0[:, 1] < 1.85)]
    src = src0 @ rot_from_axis_angle(rng.normal(size=3), 0.005).T + rng.uniform(-0.02, 0.02, 3)
    normals = O.o3d_estimate_normals(tgt, 30)
    a = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=30, icp_type="point2plane", tgt_normals=normals)
    b = O.icp(src, tgt, np.eye(4), max_corr_dist=0.1, max_iter=30, icp_type="point2plane_robust", tgt_normals=normals)
    ma = src @ a["est_transform"][:3, :3].T + a["est_transform"][:3, 3]
    mb = src @ b["est_transform"][:3, :3].T + b["est_transform"][:3, 3]
    assert a["iters"] == b["iters"] and np.abs(ma - mb).max() <= 1e-9
    # four pairs
    few_a = O.icp(src[:4], tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane", tgt_normals=normals, fixed_iters=True)
    few_b = O.icp(src[:4], tgt, np.eye(4), max_corr_dist=0.1, max_iter=1, icp_type="point2plane_robust", tgt_normals=normals, fixed_iters=True)
    assert np.array_equal(few_b["est_transform"], np.eye(4)) and not np.array_equal(few_a["est_transform"], np.eye(4))
    # the plane z = 0 with exact normals, sources 1 cm above it and tilted
    tp = np.c_[rng.uniform(0, 1, (300, 2)), np.zeros(300)]
    sp = np.c_[rng.uniform(0.1, 0.9, (200, 2)), np.zeros(200)]
    sp[:, 2] = 0.01 + 0.02 * (sp[:, 0] - 0.5)
    npl = np.tile([0.0, 0.0, 1.0], (300, 1))
    pa = O.icp(sp, tp, np.eye(4), max_corr_dist=0.05, max_iter=1, icp_type="point2plane", tgt_normals=npl, fixed_iters=True)
    pb = O.icp(sp, tp, np.eye(4), max_corr_dist=0.05, max_iter=1, icp_type="point2plane_robust", tgt_normals=npl, fixed_iters=True)
    assert np.array_equal(pb["est_transform"], np.eye(4))
    Ta = pa["est_transform"]
    assert Ta[0, 3] == 0.0 and Ta[1, 3] == 0.0 and Ta[1, 0] == 0.0  # no shift in the plane, no spin about z (sin(gamma) cos(beta))
    lifted = sp @ Ta[:3, :3].T + Ta[:3, 3]
    assert np.abs(lifted[:, 2]).max() <= 1e-5   # one step puts the tilted sheet into the plane (small-angle residue)


def _np_gicp_covariance(n, eps):
    """InitializePointCloudForGeneralizedICP's covariance from one normal, as Open3D 0.19 writes it [3P-knowledge]."""
    c = n[0]
    if c < -0.99:
        R = np.eye(3)
    else:
        v = np.cross([1.0, 0.0, 0.0], n)
        sv = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        R = np.eye(3) + sv + sv @ sv / (1 + c)
    return R @ np.diag([eps, 1.0, 1.0]) @ R.T


def _np_gicp(src, tgt, sn, tn, eps, r, max_iter, T0=None, rel=1e-6, fixed=False):
    """registration_generalized_icp [3P-knowledge] in numpy / scipy: covariances from the normals, the source's turned with
    the cloud, W = sqrtm(inv(C_q + C_s)), three rows per pair, numpy's solve of the normal equations, KD-tree search."""
    from scipy.linalg import sqrtm
    T = np.eye(4) if T0 is None else T0.copy()
    Cs = np.array([_np_gicp_covariance(n, eps) for n in sn])
    Ct = np.array([_np_gicp_covariance(n, eps) for n in tn])
    p = src @ T[:3, :3].T + T[:3, 3]
    Cs = np.einsum("ij,njk,lk->nil", T[:3, :3], Cs, T[:3, :3])
    tree = cKDTree(tgt)

    def evaluate(q):
        d, j = tree.query(q, k=1)
        ok = d * d < r * r
        return ok, j, ok.mean(), (np.sqrt((d[ok] ** 2).sum() / ok.sum()) if ok.any() else 0.0)

    ok, j, fit, rmse = evaluate(p)
    iters = 0
    for _ in range(max_iter):
        JTJ, JTr = np.zeros((6, 6)), np.zeros(6)
        for i in np.nonzero(ok)[0]:
            vs, d = p[i], p[i] - tgt[j[i]]
            W = np.real(sqrtm(np.linalg.inv(Ct[j[i]] + Cs[i])))
            A = np.c_[-np.array([[0, -vs[2], vs[1]], [vs[2], 0, -vs[0]], [-vs[1], vs[0], 0]]), np.eye(3)]
            J = W @ A
            JTJ += J.T @ J
            JTr += J.T @ (W @ d)
        U4 = np.eye(4)
        if ok.any():
            x = np.linalg.solve(JTJ, -JTr)
            al, be, ga = x[:3]
            Rz = np.array([[np.cos(ga), -np.sin(ga), 0], [np.sin(ga), np.cos(ga), 0], [0, 0, 1]])
            Ry = np.array([[np.cos(be), 0, np.sin(be)], [0, 1, 0], [-np.sin(be), 0, np.cos(be)]])
            Rx = np.array([[1, 0, 0], [0, np.cos(al), -np.sin(al)], [0, np.sin(al), np.cos(al)]])
            U4[:3, :3], U4[:3, 3] = Rz @ Ry @ Rx, x[3:]
        T = U4 @ T
        p = p @ U4[:3, :3].T + U4[:3, 3]
        Cs = np.einsum("ij,njk,lk->nil", U4[:3, :3], Cs, U4[:3, :3])
        pf, pr = fit, rmse
        ok, j, fit, rmse = evaluate(p)
        iters += 1
        if not fixed and abs(pf - fit) < rel and abs(pr - rmse) < rel:
            break
    return T, fit, rmse, iters


def test_gicp_covariances_are_open3d_s():
    """C = Rx diag(eps, 1, 1) Rx^T with Rx turning e1 onto the normal: I - (1 - eps) n n^T for a unit normal -- except for
    normals within 8 degrees of -e1 (c < -0.99), where GetRotationFromE1ToX returns the identity and the point gets e1's
    covariance whatever its normal is."""
    rng = np.random.default_rng(0)
    for eps in (1e-3, 0.0, 0.25):
        for _ in range(50):
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            C = O.gicp_covariance(n, eps)
            assert np.abs(C - _np_gicp_covariance(n, eps)).max() < 1e-15
            if n[0] >= -0.99:
                assert np.abs(C - (np.eye(3) - (1 - eps) * np.outer(n, n))).max() < 1e-13
        n = np.array([-0.995, np.sqrt(1 - 0.995 ** 2), 0.0])
        assert np.array_equal(O.gicp_covariance(n, eps), np.diag([eps, 1.0, 1.0]))


@pytest.mark.parametrize("eps", [1e-3, 0.0])
def test_gicp_oracle_walks_the_same_trajectory_as_numpy_restatement(eps):
    """utils/o3d_tools.py:40-41,51-56 (icp_type 'generalized_icp'; the reference's own epsilon is float(False) = 0): the C
    oracle against an independent numpy / scipy restatement (scipy's sqrtm and inverse, numpy's solve), from the identity
    and from a rotated start (which turns the source's covariances before the first pass)."""
    rng = np.random.default_rng(5)
    tgt = _surface(rng, 1500, noise=0.002)
    xy = rng.uniform(0.15, 1.85, (900, 2))
    src = np.c_[xy, _relief(xy)]
    src = src @ rot_from_axis_angle(rng.normal(size=3), 0.006).T + rng.uniform(-0.02, 0.02, 3)
    src, tgt = src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64)
    sn, tn = O.o3d_estimate_normals(src, 30), O.o3d_estimate_normals(tgt, 30)
    T0 = np.eye(4)
    T0[:3, :3] = rot_from_axis_angle([0.3, -1.0, 0.5], 0.004)
    T0[:3, 3] = [0.004, -0.003, 0.002]
    for init in (None, T0):
        for fixed, iters in ((True, 5), (False, 30)):
            ref = O.gicp(src, tgt, init, 0.1, iters, epsilon=eps, fixed_iters=fixed, src_normals=sn, tgt_normals=tn)
            T, fit, rmse, n_it = _np_gicp(src, tgt, sn, tn, eps, 0.1, iters, T0=init, fixed=fixed)
            assert ref["iters"] == n_it and ref["fitness"] == fit
            assert np.abs(ref["est_transform"] - T).max() < (1e-12 if eps else 1e-10)
            assert abs(ref["inlier_rmse"] - rmse) < 1e-12
            assert ref["fitness"] > 0.95
    # normals made inside the call are the ones estimate_normals() leaves (:29-30); one patch of a batch = one call
    a = O.gicp(src, tgt, None, 0.1, 30, epsilon=eps)
    b = O.gicp(src, tgt, None, 0.1, 30, epsilon=eps, src_normals=sn, tgt_normals=tn)
    assert np.array_equal(a["est_transform"], b["est_transform"])
    batch = O.piecewise_gicp(np.r_[src, src].astype(np.float32), [0, len(src), 2 * len(src)], np.r_[tgt, tgt].astype(np.float32),
                             [0, len(tgt), 2 * len(tgt)], max_corr_dist=0.1, max_iter=30, epsilon=eps)
    assert np.array_equal(batch["T"][0], a["est_transform"]) and np.array_equal(batch["T"][1], a["est_transform"])
    assert batch["iters"][0] == a["iters"]
    with pytest.raises(Exception):
        O.icp(src, tgt, None, 0.1, 30, icp_type="generalized_icp")


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
        for tag in ("generalized_icp", "generalized_icp_default"):
            if f"T_{tag}_{c}" not in g:  # (files written before the generalized estimator was dumped)
                continue
            if tag == "generalized_icp":
                assert float(g[f"epsilon_{tag}"]) == 0.0  # what utils/o3d_tools.py:41's `(False)` is read as here
            out = O.gicp(src, tgt, g[f"init_{c}"], max_corr_dist=float(g["threshold"]), max_iter=30, epsilon=float(g[f"epsilon_{tag}"]))
            T = g[f"T_{tag}_{c}"]
            moved_a = src @ out["est_transform"][:3, :3].T + out["est_transform"][:3, 3]
            assert np.abs(moved_a - (src @ T[:3, :3].T + T[:3, 3])).max() <= 1e-6, (c, tag)
            assert abs(out["fitness"] - float(g[f"fitness_{tag}_{c}"])) <= 1e-3
            assert abs(out["inlier_rmse"] - float(g[f"rmse_{tag}_{c}"])) <= 1e-5
