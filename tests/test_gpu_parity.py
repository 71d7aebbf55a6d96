"""GPU parity tests proper: every call goes through the C ABI (libf4l_hip.so) and is compared with the CPU
oracle on the same seeded inputs, or with the committed golden vectors.  Run with `-m gpu` on an MI355X.

Tolerances are those of SURVEY.md 8(d):
  Kabsch   |R - R_ref|_max <= 1e-5, |t - t_ref| <= 1e-5 max(1, |c|)
  kNN      index sets equal within exact-tie groups, d2 bit-equal (double, same operation order)
  normals  |n . n_ref| >= 1 - 1e-6
  ICP      rotation angle <= 1e-4 rad, translation / displacement <= 1e-4 m
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402
from tests._util import knn_equal_within_ties, rot_from_axis_angle, rotation_angle, same_partition  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from fusion4landslide_amd import engine
    return engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ragged(rng, sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    return off


# ------------------------------------------------------------------------------------------- Kabsch
def test_kabsch_vs_golden(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "kabsch_golden.npz"))
    names = sorted({k.rsplit("_", 1)[0] for k in g.files if k.startswith("c") and k.endswith("_R")})
    checked = 0
    for nm in names:
        src, tgt = g[nm + "_src"], g[nm + "_tgt"]
        if src.ndim != 2:
            continue
        w = g[nm + "_w"] if nm + "_w" in g.files else None
        eps, thr = float(g[nm + "_eps"]), float(g[nm + "_thr"])
        n_eff = len(src) if w is None else int((w >= thr).sum())
        if n_eff < 3:
            continue  # rank-deficient: rotation not determined (see tests/test_oracle_golden.py)
        off = np.array([0, len(src)], dtype=np.int64)
        R, t = eng.kabsch_batched(dev(src), dev(tgt), dev(off), None if w is None else dev(w), thr, eps)
        R, t = R.cpu().numpy()[0], t.cpu().numpy()[0]
        fp32 = src.dtype == np.float32
        scale = max(1.0, float(np.abs(src).max()))
        rtol, ttol = (5e-5, 2e-4 * scale) if fp32 else (1e-9, 1e-9 * scale)
        if "georef" in nm and fp32:
            rtol, ttol = 2e-2, 2.0
        assert np.abs(R - g[nm + "_R"]).max() <= rtol, nm
        assert np.abs(t - g[nm + "_t"]).max() <= ttol, nm
        checked += 1
    assert checked >= 30


def test_kabsch_ragged_batch_vs_oracle(eng):
    rng = np.random.default_rng(11)
    sizes = [3, 0, 17, 64, 65, 500, 1, 1300, 4, 255, 256, 257]
    off = ragged(rng, sizes)
    n = off[-1]
    src = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    tgt = np.empty_like(src)
    for p in range(len(sizes)):
        R0 = rot_from_axis_angle(rng.normal(size=3), rng.uniform(-0.5, 0.5))
        tgt[off[p]:off[p + 1]] = (src[off[p]:off[p + 1]] @ R0.T + rng.uniform(-1, 1, 3) + rng.normal(0, 0.01, (sizes[p], 3))).astype(np.float32)
    w = rng.uniform(0, 1, n).astype(np.float32)
    for weights, thr, eps in [(None, 0.0, 1e-7), (w, 0.0, 1e-6), (w, 0.3, 1e-6)]:
        R, t = eng.kabsch_batched(dev(src), dev(tgt), dev(off), None if weights is None else dev(weights), thr, eps)
        Rr, tr = O.kabsch_batched(src, tgt, off, weights, thr, eps)
        R, t = R.cpu().numpy(), t.cpu().numpy()
        for p, sz in enumerate(sizes):
            n_eff = sz if weights is None else int((weights[off[p]:off[p + 1]] >= thr).sum())
            if n_eff < 3:
                continue
            assert np.abs(R[p] - Rr[p]).max() <= 1e-5, (p, sz)
            assert np.abs(t[p] - tr[p]).max() <= 1e-5, (p, sz)
        assert np.allclose(R[1], np.eye(3)) and np.allclose(t[1], 0)  # empty patch -> identity
    res = eng.kabsch_residuals(dev(src), dev(tgt), dev(off), dev(Rr), dev(tr)).cpu().numpy()
    for p in range(len(sizes)):
        s, q = src[off[p]:off[p + 1]].astype(np.float64), tgt[off[p]:off[p + 1]].astype(np.float64)
        assert np.allclose(res[off[p]:off[p + 1]], np.linalg.norm(s @ Rr[p].T + tr[p] - q, axis=1), atol=1e-12)


def test_kabsch_f64_path(eng):
    rng = np.random.default_rng(12)
    src = rng.uniform(-1, 1, (200, 3)) + np.array([2647000.0, 1177000.0, 1500.0])
    R0 = rot_from_axis_angle([0.2, 0.3, 1.0], 0.02)
    c = src.mean(0)
    tgt = (src - c) @ R0.T + c + np.array([0.05, -0.02, 0.01])
    off = np.array([0, 120, 200], dtype=np.int64)
    R, t = eng.kabsch_batched(dev(src), dev(tgt), dev(off), eps=1e-6)
    for p in range(2):
        Rr, tr = O.weighted_procrustes(src[off[p]:off[p + 1]], tgt[off[p]:off[p + 1]], eps=1e-6)
        assert np.abs(R.cpu().numpy()[p] - Rr).max() < 1e-8
        # t = ct - R cs amplifies a 1e-10 rotation difference by |cs| ~ 3e6 m: compare where it matters, at the points
        s = src[off[p]:off[p + 1]]
        a = s @ R.cpu().numpy()[p].T + t.cpu().numpy()[p]
        assert np.abs(a - (s @ Rr.T + tr)).max() < 1e-7
        # NOT asserted: a == tgt.  The reference divides by (sum w + eps) (scripts/weighted_svd.py:96), which shifts
        # both centroids by ~eps/n of their magnitude: at Swiss-grid coordinates that is a ~0.4 mm misfit on exact
        # data.  The quirk is reproduced, not repaired.
        assert 1e-5 < np.abs(a - tgt[off[p]:off[p + 1]]).max() < 5e-3


# ---------------------------------------------------------------------------------------------- ICP
def _patches(n=30_000, cells=6, seed=1, origin=(0.0, 0.0, 0.0)):
    from fusion4landslide_amd import synthetic
    return synthetic.make_patches(n, cells, 1.386, seed=seed, origin=origin)


def _max_disp(d, Ta, Tb):
    worst = 0.0
    for p in range(d["P"]):
        s = d["src"][d["src_off"][p]:d["src_off"][p + 1]].astype(np.float64)
        if len(s) == 0:
            continue
        a = s @ Ta[p, :3, :3].T + Ta[p, :3, 3]
        b = s @ Tb[p, :3, :3].T + Tb[p, :3, 3]
        worst = max(worst, float(np.abs(a - b).max()))
    return worst


def _disp_per_patch(d, Ta, Tb):
    out = []
    for p in range(d["P"]):
        s = d["src"][d["src_off"][p]:d["src_off"][p + 1]].astype(np.float64)
        if len(s) == 0:
            continue
        out.append(float(np.abs((s @ Ta[p, :3, :3].T + Ta[p, :3, 3]) - (s @ Tb[p, :3, :3].T + Tb[p, :3, 3])).max()))
    return np.array(out)


@pytest.mark.parametrize("search", ["f64", "f32"])
@pytest.mark.parametrize("fixed,max_iter", [(False, 30), (True, 20)])
@pytest.mark.parametrize("origin", [(0.0, 0.0, 0.0), (2647.0, 1177.0, 1500.0)])
def test_icp_point2point_vs_oracle(eng, fixed, max_iter, origin, search):
    """search="f64" is the parity mode (Open3D arithmetic): it must reproduce the oracle's trajectory, iteration
    counts included.  search="f32" is the fast path: float32 rounding of the transformed positions (~1e-7 m) can
    flip a nearest neighbour between two almost equidistant targets; ICP then follows a different but equally
    valid trajectory on that patch, so the fast path is held to the SURVEY tolerance (1e-4 m) on >= 90 % of the
    patches with a hard cap of 2 mm, and to 1e-6 m at the median."""
    d = _patches(origin=origin)
    out = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                            max_iter=max_iter, fixed_iters=fixed, return_corr=True, search=search)
    ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=max_iter,
                          fixed_iters=fixed)
    T = out["T"].cpu().numpy()
    disp = _disp_per_patch(d, T, ref["T"])
    it = out["iters"].cpu().numpy()
    if search == "f64":
        assert disp.max() <= 1e-9
        assert np.array_equal(it, ref["iters"])
        assert np.abs(out["fitness"].cpu().numpy() - ref["fitness"]).max() == 0.0
        assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"]).max() <= 1e-10
    else:
        assert np.median(disp) <= 1e-6 and (disp <= 1e-4).mean() >= 0.9 and disp.max() <= 2e-3
        assert np.abs(out["fitness"].cpu().numpy() - ref["fitness"]).max() <= 2.5e-3
        assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"]).max() <= 1e-4
        assert (np.abs(it - ref["iters"]) <= 1).mean() > 0.8 and it.max() <= max_iter
    if fixed:
        assert (it == max_iter).all()
    # correspondence_set of patch 0 against a single-patch oracle run (same final transform => same pairs)
    s0, s1, t0, t1 = d["src_off"][0], d["src_off"][1], d["tgt_off"][0], d["tgt_off"][1]
    one = O.icp(d["src"][s0:s1], d["tgt"][t0:t1], max_corr_dist=0.1, max_iter=max_iter, fixed_iters=fixed)
    corr = out["corr"].cpu().numpy()[s0:s1]
    ref_corr = np.full(s1 - s0, -1)
    ref_corr[one["correspondence_set"][:, 0]] = one["correspondence_set"][:, 1]
    assert (corr == ref_corr).mean() > (0.9999 if search == "f64" else 0.99)


def test_icp_known_answer_planted_motion(eng):
    """Known-answer test (Open3D parity is unpinned): the target is the source cloud itself, shuffled and
    moved by a small rigid motion per patch, so the global optimum is the planted motion with zero residual and the
    first correspondences are already (mostly) the true twins.  Both ICP flavours must land on it."""
    from fusion4landslide_amd import synthetic
    rng = np.random.default_rng(5)
    c = synthetic.two_epoch_cloud(20_000, 5, 1.386, noise=0.0, seed=7, roughness=0.15)
    so, soff = synthetic.grid_partition(c["src"], 5, 1.386)
    src = c["src"][so]
    P = 25
    tgt = np.empty_like(src)
    Tt = np.tile(np.eye(4), (P, 1, 1))
    for p in range(P):
        a, b = soff[p], soff[p + 1]
        R0 = rot_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.02, 0.1)))
        cpt = src[a:b].mean(0).astype(np.float64)
        Tt[p, :3, :3] = R0
        Tt[p, :3, 3] = cpt - R0 @ cpt + rng.uniform(-0.004, 0.004, 3)
        moved = src[a:b].astype(np.float64) @ R0.T + Tt[p, :3, 3]
        tgt[a:b] = moved[rng.permutation(b - a)].astype(np.float32)
    d = dict(src=src, src_off=soff, P=P)
    start = _max_disp(d, np.tile(np.eye(4), (P, 1, 1)), Tt)
    for icp_type in ("point2point", "point2plane"):
        out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(soff), max_corr_dist=0.1, max_iter=30,
                                icp_type=icp_type, search="f64")
        T = out["T"].cpu().numpy()
        assert _max_disp(d, T, Tt) <= 2e-6 < start, icp_type  # float32 storage of the moved copy: ~1e-7 m noise
        assert (out["fitness"].cpu().numpy() == 1.0).all()
        assert out["rmse"].cpu().numpy().max() <= 1e-6
        ref = O.piecewise_icp(src, soff, tgt, soff, max_corr_dist=0.1, max_iter=30, icp_type=icp_type)
        assert _max_disp(d, T, ref["T"]) <= 1e-7


@pytest.mark.parametrize("eps", [1e-3, 0.0])
def test_generalized_icp_vs_oracle(eng, eps):
    """icp_type 'generalized_icp' (utils/o3d_tools.py:40-41,51-56; f4l_piecewise_gicp) against the oracle's restatement of
    Open3D's estimator, with Open3D's default epsilon and with the 0.0 the reference's `...ForGeneralizedICP(False)` asks for:
    the kernel forms M^-1 in closed form from the normals and never takes the matrix square root Open3D (and the oracle) take,
    so this is two routes to the same normal equations.  Patches of 40 to 3 000 points: every workgroup shape."""
    d = synthetic_patches(n=24_000, cells=5, seed=4, roughness=0.15)
    args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
    ref = O.piecewise_gicp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=30, epsilon=eps)
    out = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, icp_type="generalized_icp", gicp_epsilon=eps, return_corr=True)
    tol = 1e-7 if eps else 1e-6  # (epsilon 0: weights up to 2 / angle^2 between the two normals)
    assert _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"]).max() <= tol
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["fitness"].cpu().numpy(), ref["fitness"])
    assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"]).max() <= 1e-8
    assert np.median(ref["fitness"]) > 0.9  # (the set carries a few displaced and half-matched patches)
    # normals handed over = normals made inside the call; a fixed number of passes; a start that turns the covariances
    sn = eng.patch_normals(args[0], args[1], 30, f64=True)
    tn = eng.patch_normals(args[2], args[3], 30, f64=True)
    out2 = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, icp_type="generalized_icp", gicp_epsilon=eps,
                             src_normals=sn, tgt_normals=tn)
    assert torch.equal(out2["T"], out["T"])
    T0 = np.tile(np.eye(4), (d["P"], 1, 1))
    T0[:, :3, :3] = rot_from_axis_angle([0.3, -1.0, 0.5], 0.003)
    ref3 = O.piecewise_gicp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], init_T=T0, max_corr_dist=0.1, max_iter=4, epsilon=eps,
                            fixed_iters=True)
    out3 = eng.piecewise_icp(*args, init_T=dev(T0), max_corr_dist=0.1, max_iter=4, icp_type="generalized_icp", gicp_epsilon=eps,
                             fixed_iters=True)
    assert _disp_per_patch(d, out3["T"].cpu().numpy(), ref3["T"]).max() <= tol
    assert (out3["iters"].cpu().numpy() == 4).all()
    # small and ragged patches (one and two wavefronts, an empty source, an empty target)
    rng = np.random.default_rng(3)
    sizes = [0, 17, 64, 65, 128, 129, 300, 40]
    src_l, tgt_l = [], []
    for i, n in enumerate(sizes):
        xy = rng.uniform(0, 1.0, (max(2 * n, 60), 2))
        z = 0.2 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1]) + 0.04 * np.sin(11 * xy[:, 0]) * np.sin(9 * xy[:, 1])
        t = np.c_[xy, z]
        tgt_l.append(t if i != 7 else t[:0])
        sxy = rng.uniform(0.1, 0.9, (n, 2))
        sz = 0.2 * np.sin(3 * sxy[:, 0]) * np.cos(2 * sxy[:, 1]) + 0.04 * np.sin(11 * sxy[:, 0]) * np.sin(9 * sxy[:, 1])
        src_l.append(np.c_[sxy, sz] @ rot_from_axis_angle(rng.normal(size=3), 0.004).T + rng.uniform(-0.01, 0.01, 3))
    src = np.concatenate(src_l).astype(np.float32)
    tgt = np.concatenate(tgt_l).astype(np.float32)
    soff = np.r_[0, np.cumsum([len(x) for x in src_l])].astype(np.int64)
    toff = np.r_[0, np.cumsum([len(x) for x in tgt_l])].astype(np.int64)
    ref4 = O.piecewise_gicp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30, epsilon=eps)
    out4 = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.1, max_iter=30, icp_type="generalized_icp",
                             gicp_epsilon=eps)
    dd = dict(P=len(sizes), src=src, src_off=soff)
    assert np.array_equal(out4["iters"].cpu().numpy(), ref4["iters"])
    assert _disp_per_patch(dd, out4["T"].cpu().numpy(), ref4["T"]).max() <= 10 * tol
    assert np.array_equal(out4["fitness"].cpu().numpy(), ref4["fitness"])
    assert np.array_equal(out4["T"][0].cpu().numpy(), np.eye(4)) and np.array_equal(out4["T"][7].cpu().numpy(), np.eye(4))


@pytest.mark.parametrize("semantics", ["open3d", "robust"])
@pytest.mark.parametrize("search", ["f64", "f32"])
def test_icp_point2plane_vs_oracle(eng, search, semantics):
    """Both semantics of the point-to-plane step against their own oracle: `open3d` = F4L_ICP_P2PL_OPEN3D against the STRICT
    restatement of Open3D's step (oracle mode 1: double sums in the caller's frame, Eigen's pivoted L D L^T, applied whenever
    there is a correspondence; utils/o3d_tools.py:38-39,46-50), `robust` = the kernel's default against the oracle's robust
    variant (mode 2).  Well-posed patches: 1e-7 m (float32 normals 1e-6), and the two semantics agree with each other."""
    # metre-scale relief: a patch of a near-planar surface leaves the in-plane motion undetermined, and any two
    # solvers then differ by what they do to the null space (not a parity question)
    d = synthetic_patches(n=24_000, cells=5, seed=4, roughness=0.15)
    oracle_type = "point2plane" if semantics == "open3d" else "point2plane_robust"
    nrm = eng.patch_normals(dev(d["tgt"]), dev(d["tgt_off"]), 30)
    nrm_h = nrm.cpu().numpy().astype(np.float64)
    for p in range(d["P"]):
        t0, t1 = d["tgt_off"][p], d["tgt_off"][p + 1]
        ref_n = O.o3d_estimate_normals(d["tgt"][t0:t1].astype(np.float64), 30)
        dots = np.abs(np.sum(nrm_h[t0:t1] * ref_n, axis=1))
        assert dots.min() >= 1 - 1e-6, p
    out = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                            max_iter=30, icp_type="point2plane", tgt_normals=nrm, search=search, p2plane=semantics)
    ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=30,
                          icp_type=oracle_type)
    disp = _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"])
    if search == "f64":
        assert disp.max() <= 1e-6  # normals are handed over as float32 here, double in the oracle
        assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"]).max() <= 1e-8
    else:
        assert np.median(disp) <= 1e-6 and (disp <= 1e-4).mean() >= 0.9 and disp.max() <= 2e-3
        assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"]).max() <= 1e-4
    # normals computed inside the call are the DOUBLES Open3D's estimate_normals() leaves in the cloud (f4l_patch_normals_f64):
    # the same answer as handing those over explicitly, and -- float64 search -- closer to the oracle than with float32 normals
    nrm64 = eng.patch_normals(dev(d["tgt"]), dev(d["tgt_off"]), 30, f64=True)
    assert nrm64.dtype == torch.float64 and torch.equal(nrm64.to(torch.float32), nrm)
    out2 = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                             max_iter=30, icp_type="point2plane", search=search, p2plane=semantics)
    out3 = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                             max_iter=30, icp_type="point2plane", tgt_normals=nrm64, search=search, p2plane=semantics)
    assert torch.equal(out2["T"], out3["T"])
    if search == "f64":
        assert _disp_per_patch(d, out2["T"].cpu().numpy(), ref["T"]).max() <= 1e-7
        assert np.array_equal(out2["iters"].cpu().numpy(), ref["iters"])
        # the other semantics on the same patches: the same minimiser where the six unknowns are pinned
        other = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                                  max_iter=30, icp_type="point2plane", tgt_normals=nrm64, search=search,
                                  p2plane="robust" if semantics == "open3d" else "open3d")
        assert _disp_per_patch(d, other["T"].cpu().numpy(), out3["T"].cpu().numpy()).max() <= 1e-7
    if search == "f64" and semantics == "robust":
        # the L D L^T solve against the pivoted elimination it replaced (F4L_ICP_DEBUG bit 256)
        os.environ["F4L_ICP_DEBUG"] = "256"
        try:
            out4 = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                                     max_iter=30, icp_type="point2plane", tgt_normals=nrm64, search=search)
        finally:
            del os.environ["F4L_ICP_DEBUG"]
        assert torch.equal(out4["iters"], out3["iters"])
        assert _disp_per_patch(d, out4["T"].cpu().numpy(), out3["T"].cpu().numpy()).max() <= 1e-11
    with pytest.raises(ValueError):
        eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), icp_type="generalized")
    with pytest.raises(ValueError):
        eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), icp_type="point2plane", p2plane="eigen")


def test_generalized_icp_with_a_singular_pair_says_so(eng):
    """generalized ICP at the reference's epsilon = 0 on an exactly planar patch (both normals (0, 0, 1): the pair
    covariance M is singular, 1 / det = inf).  Open3D returns a NaN transform; the kernel keeps its last finite transform,
    stops and flags the patch iters = -2 (include/f4l.h) -- and a well-posed patch of the same launch is untouched.  With
    epsilon > 0 the same planar patch iterates normally."""
    rng = np.random.default_rng(5)
    tgt_a = np.c_[rng.uniform(0, 1, (300, 2)), np.zeros(300)]
    src_a = np.c_[rng.uniform(0.1, 0.9, (200, 2)), np.full(200, 0.01)]
    d = synthetic_patches(n=3_000, cells=1, seed=4, roughness=0.15)
    src = np.r_[src_a, d["src"]].astype(np.float32)
    tgt = np.r_[tgt_a, d["tgt"]].astype(np.float32)
    soff = np.array([0, len(src_a), len(src)], np.int64)
    toff = np.array([0, len(tgt_a), len(tgt)], np.int64)
    nt = np.r_[np.tile([0.0, 0.0, 1.0], (len(tgt_a), 1)), O.o3d_estimate_normals(d["tgt"].astype(np.float64), 30)]
    ns = np.r_[np.tile([0.0, 0.0, 1.0], (len(src_a), 1)), O.o3d_estimate_normals(d["src"].astype(np.float64), 30)]
    args = (dev(src), dev(soff), dev(tgt), dev(toff))
    out = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, icp_type="generalized_icp", gicp_epsilon=0.0,
                            tgt_normals=dev(nt), src_normals=dev(ns))
    it = out["iters"].cpu().numpy()
    T = out["T"].cpu().numpy()
    assert it[0] == -2 and np.array_equal(T[0], np.eye(4))       # the first step already is not finite: the start stays
    assert it[1] > 0 and np.isfinite(T).all()
    ok = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, icp_type="generalized_icp", gicp_epsilon=1e-3,
                           tgt_normals=dev(nt), src_normals=dev(ns))
    assert (ok["iters"] > 0).all()
    assert abs(ok["T"][0, 2, 3].item() + 0.01) < 1e-5            # the sheet is lifted onto the plane


def test_icp_point2plane_where_the_two_semantics_part(eng):
    """Where Open3D's step and the robust one differ (include/f4l.h, F4L_ICP_P2PL_OPEN3D): (a) a patch left with four
    correspondences -- robust: no step; Open3D (kernel and strict oracle alike): the singular system's "solution" is applied,
    the patch moves, and since that solution is rounding noise the two sides need not agree on where to; (b) an exactly planar
    target with normals (0, 0, 1) near the origin -- robust: no step (rank 3); Open3D: Eigen's pseudo-inverse of D moves the
    sheet along what the data sees (tilt, lift) and not at all along what it does not: kernel = strict oracle to 1e-9 m."""
    rng = np.random.default_rng(12)
    gx, gy = np.meshgrid(np.arange(8) * 0.1, np.arange(8) * 0.1)
    tgt_a = np.c_[gx.ravel(), gy.ravel(), 0.05 * np.sin(3 * gx.ravel()) * np.cos(2 * gy.ravel())]
    near = tgt_a[[9, 20, 35, 50]] + rng.normal(0, 0.004, (4, 3))
    far = tgt_a[:7] + np.array([0.0, 0.0, 1.0])
    src_a = np.r_[near, far]
    tgt_b = np.c_[rng.uniform(0, 1, (300, 2)), np.zeros(300)]
    src_b = np.c_[rng.uniform(0.1, 0.9, (200, 2)), np.zeros(200)]
    src_b[:, 2] = 0.01 + 0.02 * (src_b[:, 0] - 0.5)
    src = np.r_[src_a, src_b].astype(np.float32)
    tgt = np.r_[tgt_a, tgt_b].astype(np.float32)
    soff, toff = np.array([0, len(src_a), len(src)], np.int64), np.array([0, len(tgt_a), len(tgt)], np.int64)
    nrm = np.r_[O.o3d_estimate_normals(tgt[:len(tgt_a)].astype(np.float64), 30), np.tile([0.0, 0.0, 1.0], (len(tgt_b), 1))]
    res = {}
    for sem in ("robust", "open3d"):
        out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.05, max_iter=1, fixed_iters=True,
                                icp_type="point2plane", tgt_normals=dev(nrm), p2plane=sem)
        res[sem] = out["T"].cpu().numpy()
    assert np.array_equal(res["robust"][0], np.eye(4)) and np.array_equal(res["robust"][1], np.eye(4))
    assert not np.array_equal(res["open3d"][0], np.eye(4)) and np.isfinite(res["open3d"]).all()
    ref_a = O.icp(src[:len(src_a)].astype(np.float64), tgt[:len(tgt_a)].astype(np.float64), max_corr_dist=0.05, max_iter=1,
                  fixed_iters=True, icp_type="point2plane", tgt_normals=nrm[:len(tgt_a)])
    assert not np.array_equal(ref_a["est_transform"], np.eye(4))
    ref_b = O.icp(src[len(src_a):].astype(np.float64), tgt[len(tgt_a):].astype(np.float64), max_corr_dist=0.05, max_iter=1,
                  fixed_iters=True, icp_type="point2plane", tgt_normals=nrm[len(tgt_a):])
    Tb, Rb = res["open3d"][1], ref_b["est_transform"]
    sb = src[len(src_a):].astype(np.float64)
    assert np.abs((sb @ Tb[:3, :3].T + Tb[:3, 3]) - (sb @ Rb[:3, :3].T + Rb[:3, 3])).max() <= 1e-9
    assert Tb[0, 3] == 0.0 and Tb[1, 3] == 0.0 and Tb[1, 0] == 0.0   # nothing along what the data does not see
    assert np.abs((sb @ Tb[:3, :3].T + Tb[:3, 3])[:, 2]).max() <= 1e-5


def test_icp_point2plane_refuses_a_singular_step(eng):
    """The ROBUST semantics (the batched calls' default; Open3D's own are F4L_ICP_P2PL_OPEN3D, tested above): a point-to-plane
    step that cannot pin its six unknowns is not taken: fewer than six correspondences (the robust oracle's rule too), or an
    exactly planar target with parallel normals (rank 3: pivots below 1e-13 of the diagonal).  The transform stays
    where it was, fitness and rmse are those of the start, and the loop ends on its criteria -- where solving the singular
    system threw a four-pair patch 55 m (tools/gpu/fuzz_icp.py 1 2250095 f64 n32)."""
    rng = np.random.default_rng(12)
    gx, gy = np.meshgrid(np.arange(8) * 0.1, np.arange(8) * 0.1)
    tgt_a = np.c_[gx.ravel(), gy.ravel(), 0.05 * np.sin(3 * gx.ravel()) * np.cos(2 * gy.ravel())]
    near = tgt_a[[9, 20, 35, 50]] + rng.normal(0, 0.004, (4, 3))
    far = tgt_a[:7] + np.array([0.0, 0.0, 1.0])
    src_a = np.r_[near, far]
    tgt_b = np.c_[rng.uniform(0, 1, (300, 2)), np.zeros(300)]  # a plane, normals exactly (0, 0, 1)
    src_b = np.c_[rng.uniform(0.1, 0.9, (200, 2)), np.full(200, 0.01)]
    origin = np.array([2647.0, 1177.0, 1500.0])
    for shift in (np.zeros(3), origin):
        src = (np.r_[src_a, src_b] + shift).astype(np.float32)
        tgt = (np.r_[tgt_a, tgt_b] + shift).astype(np.float32)
        soff, toff = np.array([0, len(src_a), len(src)], np.int64), np.array([0, len(tgt_a), len(tgt)], np.int64)
        nrm = np.r_[O.o3d_estimate_normals(tgt[:len(tgt_a)].astype(np.float64), 30), np.tile([0.0, 0.0, 1.0], (len(tgt_b), 1))]
        out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.05, max_iter=30, icp_type="point2plane",
                                tgt_normals=dev(nrm))
        T = out["T"].cpu().numpy()
        assert np.array_equal(T[0], np.eye(4)) and np.array_equal(T[1], np.eye(4)), T
        fit = out["fitness"].cpu().numpy()
        assert abs(fit[0] - 4 / 11) < 1e-12 and fit[1] > 0.8  # (the plane: nearly every point matched, and still rank 3)
        ref = O.icp(src[:len(src_a)].astype(np.float64), tgt[:len(tgt_a)].astype(np.float64), max_corr_dist=0.05, max_iter=30,
                    icp_type="point2plane_robust", tgt_normals=nrm[:len(tgt_a)])
        assert np.array_equal(ref["est_transform"], np.eye(4)) and abs(ref["fitness"] - 4 / 11) < 1e-12
        assert abs(out["rmse"].cpu().numpy()[0] - ref["inlier_rmse"]) < 1e-9


def test_patch_normals_lane_per_query_equals_wave_per_query(eng):
    """f4l_patch_normals: the lane-per-query kernel (patches up to 8192 points, k <= 36) against the wave-per-query kernel it
    replaces there (F4L_PATCH_NORMALS_WAVES): the same neighbour lists, hence the same normals bit for bit -- patches of every
    size (fewer points than k, one point, empty, beyond the lane kernel's limit), a lattice (exact distance ties), duplicated
    points, a collinear patch (degenerate bounding box), georeferenced coordinates."""
    import os
    rng = np.random.default_rng(9)
    parts = []
    for n in (0, 1, 2, 5, 29, 30, 31, 64, 257, 500, 1500, 9000):
        xy = rng.uniform(0, 1, (n, 2)) * max(n, 1) ** 0.5 * 0.05
        parts.append(np.c_[xy, 0.1 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1]) + rng.normal(0, 0.002, n)])
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3), indexing="ij"), -1).reshape(-1, 3) * 0.05
    parts.append(g)                                                        # lattice: ties
    parts.append(np.repeat(parts[9][:120], 3, axis=0))                     # every point three times
    parts.append(np.c_[np.linspace(0, 1, 200), np.zeros(200), np.zeros(200)])  # collinear
    parts.append(parts[9] + np.array([2.6e6, 1.2e6, 1800.0]))              # georeferenced
    pts = np.concatenate(parts).astype(np.float32)
    off = np.concatenate([[0], np.cumsum([len(a) for a in parts])]).astype(np.int64)
    for k in (30, 8, 36):
        fast = eng.patch_normals(dev(pts), dev(off), k, f64=True).cpu().numpy()
        os.environ["F4L_PATCH_NORMALS_WAVES"] = "1"
        try:
            slow = eng.patch_normals(dev(pts), dev(off), k, f64=True).cpu().numpy()
        finally:
            del os.environ["F4L_PATCH_NORMALS_WAVES"]
        # round 6: every kernel and path sums the covariance in the order of the neighbour list (Open3D's ComputeCovariance), so the
        # same neighbours give the same BITS -- the lattice and the duplicates send the lane kernel through its overflow path
        assert np.array_equal(fast, slow), (k, int((fast != slow).any(axis=1).sum()), np.abs(fast - slow).max())
        assert np.isfinite(fast).all() and np.abs(np.linalg.norm(fast, axis=1) - 1).max() < 1e-12
        f32 = eng.patch_normals(dev(pts), dev(off), k).cpu().numpy()
        assert np.array_equal(f32, fast.astype(np.float32))


def synthetic_patches(**kw):
    from fusion4landslide_amd import synthetic
    return synthetic.make_patches(kw.pop("n"), kw.pop("cells"), 1.386, **kw)


def test_icp_edge_cases(eng):
    rng = np.random.default_rng(2)
    # patch 0: empty source; 1: empty target; 2: single points in range; 3: nothing within range; 4: normal
    base = rng.uniform(0, 1, (300, 3)).astype(np.float32)
    base[:, 2] *= 0.05
    src_list = [np.zeros((0, 3), np.float32), base[:10], base[:1], base[:20], base[:150]]
    tgt_list = [base[:10], np.zeros((0, 3), np.float32), base[:1] + np.float32(0.01), base[:20] + np.float32(5.0),
                (base[150:] + np.float32(0.004))]
    src, tgt = np.concatenate(src_list), np.concatenate(tgt_list)
    soff, toff = ragged(rng, [len(a) for a in src_list]), ragged(rng, [len(a) for a in tgt_list])
    out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.1, max_iter=30, return_corr=True)
    ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    T, fit = out["T"].cpu().numpy(), out["fitness"].cpu().numpy()
    for p in (0, 1, 3):
        assert np.allclose(T[p], np.eye(4)) and fit[p] == 0.0 and out["rmse"].cpu().numpy()[p] == 0.0
    assert np.allclose(T[2][:3, 3], [0.01, 0.01, 0.01], atol=1e-6) and fit[2] == 1.0
    assert np.abs(T - ref["T"]).max() < 1e-4
    assert np.allclose(fit, ref["fitness"], atol=1e-2)
    corr = out["corr"].cpu().numpy()
    assert (corr[soff[1]:soff[2]] == -1).all() and (corr[soff[3]:soff[4]] == -1).all()
    # max_corr_dist <= 0: Open3D returns the init untouched
    T0 = np.tile(np.eye(4), (5, 1, 1))
    T0[:, 0, 3] = 0.5
    out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), init_T=dev(T0), max_corr_dist=0.0)
    assert np.allclose(out["T"].cpu().numpy(), T0)


def test_icp_large_patch_global_path_matches_lds_path(eng):
    # one patch beyond the LDS budget (8192 target points) must agree with the oracle too
    rng = np.random.default_rng(8)
    n = 9000
    xy = rng.uniform(0, 3, (n, 2))
    tgt = np.c_[xy, 0.2 * np.sin(2 * xy[:, 0]) * np.cos(3 * xy[:, 1])].astype(np.float32)
    xy2 = rng.uniform(0.2, 2.8, (3000, 2))
    src0 = np.c_[xy2, 0.2 * np.sin(2 * xy2[:, 0]) * np.cos(3 * xy2[:, 1])]
    R0 = rot_from_axis_angle([0, 0.2, 1], 0.004)
    src = (src0 @ R0.T + np.array([0.01, -0.015, 0.008])).astype(np.float32)
    soff, toff = np.array([0, 3000], np.int64), np.array([0, n], np.int64)
    ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    d = dict(src=src, src_off=soff, P=1)
    for search, tol in (("f64", 1e-9), ("f32", 1e-4)):
        out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.1, max_iter=30, search=search)
        assert _max_disp(d, out["T"].cpu().numpy(), ref["T"]) <= tol, search


@pytest.mark.parametrize("waves", ["1", "2", "4"])
def test_icp_every_workgroup_shape_matches_oracle(eng, waves, monkeypatch):
    """The host picks 1, 2 or 4 waves per patch from the patch count; force each shape on the same input (float64
    search: must reproduce the oracle's trajectory whatever the shape) and on the fast path (same statistics)."""
    monkeypatch.setenv("F4L_ICP_WAVES", waves)
    d = _patches(n=20_000, cells=5, seed=9)
    ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=30)
    out = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                            max_iter=30, search="f64")
    assert _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"]).max() <= 1e-9
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    out32 = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1,
                              max_iter=30, search="f32")
    disp = _disp_per_patch(d, out32["T"].cpu().numpy(), ref["T"])
    assert np.median(disp) <= 1e-5 and (disp <= 1e-4).mean() >= 0.85 and disp.max() <= 2e-3


def test_icp_lane_group_kernel_never_sees_a_patch_beyond_its_capacity(eng, monkeypatch):
    """icp_rows.h holds at most 4 * LP points per patch.  A caller that UNDERSTATES max_src_patch / max_tgt_patch
    (the launch is planned from those two numbers) must get the same transforms as one that states them exactly: patches
    larger than stated go to icp_kernel through a class of their own (icp_launch_host), never into the lane-group kernel,
    whose LDS they would overrun."""
    rng = np.random.default_rng(77)
    sizes = np.r_[rng.integers(20, 60, 120), [300, 200, 129, 65, 90, 500]]
    rng.shuffle(sizes)
    src_l, tgt_l = [], []
    for m in sizes:
        m = int(m)
        side = max(0.15, np.sqrt(m / 120.0))
        xy = rng.uniform(0, side, (m + 3, 2))
        t = np.c_[xy, 0.2 * np.sin(2.3 * xy[:, 0] / side) * np.cos(1.9 * xy[:, 1] / side)]
        xy2 = rng.uniform(0.05 * side, 0.95 * side, (m, 2))
        s = np.c_[xy2, 0.2 * np.sin(2.3 * xy2[:, 0] / side) * np.cos(1.9 * xy2[:, 1] / side)]
        s = s @ rot_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.008)).T + rng.uniform(-0.02, 0.02, 3)
        o = rng.uniform(0, 40, 3)
        src_l.append(s + o); tgt_l.append(t + o)
    src, tgt = np.concatenate(src_l).astype(np.float32), np.concatenate(tgt_l).astype(np.float32)
    soff, toff = ragged(None, [len(a) for a in src_l]), ragged(None, [len(a) for a in tgt_l])
    args = (dev(src), dev(soff), dev(tgt), dev(toff))
    monkeypatch.setenv("F4L_ICP_ROWS", "0")
    exact = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
    monkeypatch.setenv("F4L_ICP_ROWS", "1")
    for stated in (60, 64, 100, 128):   # one lane width, the other, and both class layouts (with / without the small classes)
        for small in ("0", "1"):
            monkeypatch.setenv("F4L_ICP_SMALLCLASSES", small) if small == "1" else monkeypatch.delenv("F4L_ICP_SMALLCLASSES", raising=False)
            out = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True,
                                    max_src_patch=stated, max_tgt_patch=stated)
            assert (out["iters"] >= 0).all()
            # (the two kernels sum in different orders: the oracle's trajectory to 1e-9 m either way, not the same bits)
            well = (exact["fitness"] > 0.8).cpu().numpy()
            assert well.sum() > 100
            Ta, Tb = out["T"].cpu().numpy(), exact["T"].cpu().numpy()
            for p in np.flatnonzero(well):
                sp = src[soff[p]:soff[p + 1]].astype(np.float64)
                assert np.abs((sp @ Ta[p, :3, :3].T + Ta[p, :3, 3]) - (sp @ Tb[p, :3, :3].T + Tb[p, :3, 3])).max() <= 1e-9, (p, stated, small)
            wt = torch.from_numpy(well).cuda()
            assert torch.equal(out["iters"][wt], exact["iters"][wt]) and torch.equal(out["fitness"][wt], exact["fitness"][wt])
            pt_well = torch.repeat_interleave(wt, torch.from_numpy(np.diff(soff)).cuda())
            assert torch.equal(out["corr"][pt_well], exact["corr"][pt_well]), (stated, small)


@pytest.mark.parametrize("fused", [False, True])
def test_icp_several_small_patches_per_wave_match_oracle(eng, fused, monkeypatch):
    """icp_rows.h (F4L_ICP_ROWS=1; by itself it runs for the float32 search from 64 k patches on): patches of up to 64 points on
    16 lanes, up to 128 on 32, four / two per wave, no workgroup barrier.  Supervoxel-sized patches of every size from empty to
    128 points (uneven: both lane widths and the ordinary kernel for the larger ones run side by side), float64 search against
    the oracle's trajectories (1e-9 m on well-posed patches, equal iteration counts, fitness, final correspondences), the float32
    search against its own bounds, and the fused launch (Kabsch start from correspondences, rows) against the three calls."""
    rng = np.random.default_rng(41)
    sizes = np.r_[rng.integers(20, 64, 150), rng.integers(64, 128, 100), rng.integers(128, 300, 12), [0, 1, 2, 3, 5, 64, 128, 63, 65]]
    src_l, tgt_l = [], []
    for m in sizes:
        m = int(m)
        side = max(0.15, np.sqrt(max(m, 1) / 120.0))
        mt = max(0, m + int(rng.integers(-4, 5))) if m > 3 else m
        mt = min(mt, 64) if m <= 64 else (min(mt, 128) if m <= 128 else mt)
        xy = rng.uniform(0, side, (mt, 2))
        t = np.c_[xy, 0.2 * np.sin(2.3 * xy[:, 0] / side) * np.cos(1.9 * xy[:, 1] / side) + rng.normal(0, 0.002, mt)]
        xy2 = rng.uniform(0.05 * side, 0.95 * side, (m, 2))
        s = np.c_[xy2, 0.2 * np.sin(2.3 * xy2[:, 0] / side) * np.cos(1.9 * xy2[:, 1] / side)]
        s = s @ rot_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.008)).T + rng.uniform(-0.02, 0.02, 3)
        o = rng.uniform(0, 40, 3)
        src_l.append(s + o); tgt_l.append(t + o)
    src, tgt = np.concatenate(src_l).astype(np.float32), np.concatenate(tgt_l).astype(np.float32)
    soff, toff = ragged(None, [len(a) for a in src_l]), ragged(None, [len(a) for a in tgt_l])
    d = dict(src=src, src_off=soff, P=len(sizes))

    def _disp_per_patch(dd, Ta, Tb):  # (like the module's, with a 0 for an empty patch: indexable by patch masks)
        out = np.zeros(dd["P"])
        for p in range(dd["P"]):
            s = dd["src"][dd["src_off"][p]:dd["src_off"][p + 1]].astype(np.float64)
            if len(s):
                out[p] = np.abs((s @ Ta[p, :3, :3].T + Ta[p, :3, 3]) - (s @ Tb[p, :3, :3].T + Tb[p, :3, 3])).max()
        return out
    args = (dev(src), dev(soff), dev(tgt), dev(toff))
    monkeypatch.setenv("F4L_ICP_ROWS", "1")
    monkeypatch.setenv("F4L_ICP_SMALLCLASSES", "1")
    if not fused:
        ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
        out = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
        well = (ref["fitness"] > 0.8) & (np.diff(soff) >= 12)
        assert well.sum() > 200
        assert _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"])[well].max() <= 1e-9
        assert np.array_equal(out["iters"].cpu().numpy()[well], ref["iters"][well])
        assert np.array_equal(out["fitness"].cpu().numpy()[well], ref["fitness"][well])
        assert np.abs(out["rmse"].cpu().numpy() - ref["rmse"])[well].max() <= 1e-10
        monkeypatch.setenv("F4L_ICP_ROWS", "0")
        plain = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
        monkeypatch.setenv("F4L_ICP_ROWS", "1")
        wt = torch.from_numpy(well).cuda()
        pt_well = torch.repeat_interleave(wt, torch.from_numpy(np.diff(soff)).cuda())
        assert torch.equal(out["corr"][pt_well], plain["corr"][pt_well])
        assert (out["iters"][np.diff(soff) == 0] == 0).all()  # an empty source patch: nothing to iterate on
        out32 = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f32")
        disp = _disp_per_patch(d, out32["T"].cpu().numpy(), ref["T"])[well]
        assert np.median(disp) <= 1e-5 and (disp <= 1e-4).mean() >= 0.85 and disp.max() <= 2e-3
        again = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
        for key in ("T", "fitness", "rmse", "iters", "corr"):
            assert torch.equal(out[key], again[key]), key   # run-to-run bit identical
    else:
        cs, ct = dev(src), dev(src.astype(np.float64) + 0.003)  # Kabsch input: every source point and a shifted copy
        ct = ct.to(torch.float32)
        loop = eng.patch_loop(*args, cs, ct, dev(soff), None, 0.0, 1e-6, max_corr_dist=0.1, max_iter=20, fixed_iters=True,
                              search="f64", min_corr=3)
        T0 = eng.kabsch_transforms(cs, ct, dev(soff), None, 0.0, 1e-6)
        monkeypatch.setenv("F4L_ICP_ROWS", "0")
        sep = eng.piecewise_icp(*args, init_T=T0, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search="f64")
        n_corr = np.diff(soff)
        run = torch.from_numpy(n_corr >= 3).cuda()
        well = ((sep["fitness"] > 0.8) & run).cpu().numpy()
        assert _disp_per_patch(d, loop["T"].cpu().numpy(), sep["T"].cpu().numpy())[well].max() <= 1e-9
        assert torch.equal(loop["iters"][~run], torch.full_like(loop["iters"][~run], -1))   # fewer than min_corr pairs: skipped
        rows = eng.apply_transform(dev(src), dev(soff), loop["T"])
        pt_run = torch.repeat_interleave(run, torch.from_numpy(n_corr).cuda())
        assert torch.equal(loop["rows"][pt_run], rows[pt_run]) and bool((loop["rows"][~pt_run] == 0).all())


@pytest.mark.parametrize("switch,what", [("4", "no certificates: every point searched in every pass"),
                                         ("8", "no bound from the previous correspondence"),
                                         ("128", "Jacobi SVD instead of Newton on SO(3)")])
def test_icp_shortcuts_do_not_change_the_answer(eng, switch, what, monkeypatch):
    """The certificates, the bounded search and the Newton solve are shortcuts, not approximations: switching each off
    (F4L_ICP_DEBUG bits) must give the same transforms (float64 search: to 1e-9 m with equal iteration counts and
    correspondences)."""
    d = _patches(n=30_000, cells=6, seed=14)
    args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
    kw = dict(max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
    base = eng.piecewise_icp(*args, **kw)
    monkeypatch.setenv("F4L_ICP_DEBUG", switch)
    alt = eng.piecewise_icp(*args, **kw)
    monkeypatch.delenv("F4L_ICP_DEBUG")
    assert _disp_per_patch(d, alt["T"].cpu().numpy(), base["T"].cpu().numpy()).max() <= 1e-9, what
    assert torch.equal(alt["iters"], base["iters"]) and torch.equal(alt["corr"], base["corr"]), what
    assert (alt["fitness"] - base["fitness"]).abs().max().item() == 0.0


@pytest.mark.parametrize("search", ["f32", "f64"])
def test_icp_is_bit_reproducible_run_to_run(eng, search):
    """Counting sort with atomics, per-wave queues, rotating solver wave: none of it may leak scheduling order into the
    results.  Same input, eight launches, bit-identical outputs (tools/gpu/determinism.py is the long version)."""
    d = _patches(n=60_000, cells=8, seed=15)
    args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
    first = None
    for _ in range(8):
        out = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search=search, return_corr=True)
        sig = (out["T"], out["fitness"], out["rmse"], out["iters"], out["corr"])
        if first is None:
            first = [t.clone() for t in sig]
        else:
            assert all(torch.equal(a, b) for a, b in zip(sig, first))


def test_certificates_are_exact_in_float32_arithmetic(eng, monkeypatch):
    """One update after the first evaluation: both runs share pass 0 bit for bit (everything is searched), so the
    transform of pass 1 is identical and the correspondences of pass 1 -- certified in one run, searched in the other --
    must be identical too, in the float32 search as well.  (Longer runs cannot be compared bitwise in float32: the two
    paths add the same terms in a different order.)"""
    rng = np.random.default_rng(16)
    d = _patches(n=40_000, cells=6, seed=16)
    P = d["P"]
    T0 = np.tile(np.eye(4), (P, 1, 1))
    for p in range(P):
        T0[p, :3, :3] = rot_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.01))
        T0[p, :3, 3] = rng.uniform(-0.03, 0.03, 3)
    args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
    for search in ("f32", "f64"):
        kw = dict(init_T=dev(T0), max_corr_dist=0.1, max_iter=1, fixed_iters=True, search=search, return_corr=True)
        with_cert = eng.piecewise_icp(*args, **kw)
        monkeypatch.setenv("F4L_ICP_DEBUG", "4")
        without = eng.piecewise_icp(*args, **kw)
        monkeypatch.delenv("F4L_ICP_DEBUG")
        assert torch.equal(with_cert["corr"], without["corr"]), search
        assert torch.equal(with_cert["fitness"], without["fitness"]), search
        # same pairs, summed in a different order (float32: centred float32 partial sums; patches with a handful of badly
        # placed correspondences amplify that rounding)
        disp = _disp_per_patch(d, with_cert["T"].cpu().numpy(), without["T"].cpu().numpy())
        assert np.median(disp) <= (1e-7 if search == "f32" else 1e-12) and disp.max() <= (2e-3 if search == "f32" else 1e-9)


def test_icp_medium_patches_without_room_for_every_lds_array(eng):
    """Patches of a few thousand points: the LDS plan drops the staged sources (and, beyond, the certificate arrays)
    before it gives up the grid; results must not depend on which arrays made it into LDS."""
    rng = np.random.default_rng(21)
    P, n = 3, 6000
    src_l, tgt_l = [], []
    for p in range(P):
        xy = rng.uniform(0, 4, (n, 2))
        tgt_l.append(np.c_[xy, 0.3 * np.sin(1.7 * xy[:, 0]) * np.cos(2.3 * xy[:, 1]) + rng.normal(0, 0.003, n)])
        xy2 = rng.uniform(0.3, 3.7, (n - 500 * p, 2))
        s = np.c_[xy2, 0.3 * np.sin(1.7 * xy2[:, 0]) * np.cos(2.3 * xy2[:, 1])]
        R0 = rot_from_axis_angle(rng.normal(size=3), 0.003)
        src_l.append(s @ R0.T + rng.uniform(-0.02, 0.02, 3))
    src, tgt = np.concatenate(src_l).astype(np.float32), np.concatenate(tgt_l).astype(np.float32)
    soff, toff = ragged(rng, [len(a) for a in src_l]), ragged(rng, [len(a) for a in tgt_l])
    ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    d = dict(src=src, src_off=soff, P=P)
    out = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.1, max_iter=30, search="f64",
                            return_corr=True)
    assert _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"]).max() <= 1e-9
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.abs(out["fitness"].cpu().numpy() - ref["fitness"]).max() == 0.0
    out32 = eng.piecewise_icp(dev(src), dev(soff), dev(tgt), dev(toff), max_corr_dist=0.1, max_iter=30, search="f32")
    assert _disp_per_patch(d, out32["T"].cpu().numpy(), ref["T"]).max() <= 1e-4


@pytest.mark.parametrize("search", ["f64", "f32"])
def test_icp_dense_patches_on_a_grid_finer_than_the_radius(eng, search, monkeypatch):
    """Patches as wide as max_corr_dist itself with hundreds of points (BASELINE config C3's regime): the per-patch
    grid is subdivided, searches go through the wide stencil, and pass 0 is preceded by the narrow look-up around each
    point's own cell.  Same answers as the oracle, and the narrow look-up (F4L_ICP_DEBUG bit 16 switches it off) is a
    shortcut, not an approximation."""
    from fusion4landslide_amd import synthetic
    d = synthetic.make_patches(40_000, 9, 0.1, seed=23)
    assert d["max_tgt"] > 400
    args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
    kw = dict(max_corr_dist=0.1, max_iter=30, search=search, return_corr=True)
    out = eng.piecewise_icp(*args, **kw)
    monkeypatch.setenv("F4L_ICP_DEBUG", "16")
    alt = eng.piecewise_icp(*args, **kw)
    monkeypatch.delenv("F4L_ICP_DEBUG")
    assert torch.equal(alt["corr"], out["corr"]) and torch.equal(alt["iters"], out["iters"])
    # (same correspondences, summed in another order: transforms agree to rounding, not to the bit)
    same = _disp_per_patch(d, alt["T"].cpu().numpy(), out["T"].cpu().numpy())
    assert same.max() <= (1e-9 if search == "f64" else 1e-4)
    ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=30)
    disp = _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"])
    if search == "f64":
        assert disp.max() <= 1e-9
        assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        assert np.abs(out["fitness"].cpu().numpy() - ref["fitness"]).max() == 0.0
    else:
        assert np.median(disp) <= 1e-5 and (disp <= 1e-4).mean() >= 0.85


@pytest.mark.parametrize("mix", ["few huge patches among small ones", "many patches of one or two wavefronts"])
def test_icp_size_classes_match_single_launch_and_oracle(eng, mix, monkeypatch):
    """Patches of very different sizes: the host bins them by size on the device and launches each class with its own
    LDS plan and workgroup shape (F4L_ICP_NOCLASSES=1: one launch sized for the largest patch).  Same correspondences
    and iteration counts either way, and the oracle's transforms."""
    rng = np.random.default_rng(31)
    if mix.startswith("few"):
        sizes = [5000, 150, 0, 300, 2500] + [int(v) for v in rng.integers(40, 700, 75)]
    else:  # >= 512 patches, mean well below the largest: classes of 64, 128 and the rest
        sizes = [0, 1, 2, 3] + [int(v) for v in rng.integers(5, 110, 560)] + [int(v) for v in rng.integers(110, 400, 40)]
    src_l, tgt_l = [], []
    for n in sizes:
        side = max(0.2, np.sqrt(n / 400.0))  # ~400 points per square metre
        xy = rng.uniform(0, side, (n, 2))
        tgt_l.append(np.c_[xy, 0.3 * np.sin(1.7 * xy[:, 0]) * np.cos(2.3 * xy[:, 1]) + rng.normal(0, 0.003, n)])
        m = max(0, n - int(rng.integers(0, 30)))
        xy2 = rng.uniform(0.05 * side, 0.95 * side, (m, 2))
        sp = np.c_[xy2, 0.3 * np.sin(1.7 * xy2[:, 0]) * np.cos(2.3 * xy2[:, 1])]
        R0 = rot_from_axis_angle(rng.normal(size=3), 0.003)
        src_l.append(sp @ R0.T + rng.uniform(-0.02, 0.02, 3))
    src, tgt = np.concatenate(src_l).astype(np.float32), np.concatenate(tgt_l).astype(np.float32)
    soff, toff = ragged(rng, [len(a) for a in src_l]), ragged(rng, [len(a) for a in tgt_l])
    args = (dev(src), dev(soff), dev(tgt), dev(toff))
    kw = dict(max_corr_dist=0.1, max_iter=30, search="f64", return_corr=True)
    out = eng.piecewise_icp(*args, **kw)
    monkeypatch.setenv("F4L_ICP_NOCLASSES", "1")
    one = eng.piecewise_icp(*args, **kw)
    monkeypatch.delenv("F4L_ICP_NOCLASSES")
    assert torch.equal(one["corr"], out["corr"]) and torch.equal(one["iters"], out["iters"])
    d = dict(src=src, src_off=soff, P=len(sizes))
    assert _disp_per_patch(d, one["T"].cpu().numpy(), out["T"].cpu().numpy()).max() <= 1e-9
    ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=30)
    assert _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"]).max() <= 1e-9
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.abs(out["fitness"].cpu().numpy() - ref["fitness"]).max() == 0.0
    out32 = eng.piecewise_icp(*args, max_corr_dist=0.1, max_iter=30, search="f32")
    disp = _disp_per_patch(d, out32["T"].cpu().numpy(), ref["T"])
    assert np.median(disp) <= 1e-5 and (disp <= 1e-4).mean() >= 0.85


@pytest.mark.parametrize("search,icp_type", [("f64", "point2point"), ("f32", "point2point"), ("f64", "point2plane")])
def test_icp_throughput_shape_equals_the_latency_shape(eng, search, icp_type, monkeypatch):
    """Batches of tens of thousands of patches run as two-wave workgroups with the register budget of three waves per SIMD
    (float64) and a size class of their own for the bulk of the patches (icp_launch_host: `throughput`); smaller batches as
    four-wave workgroups.  Forced onto a small uneven batch, the throughput shape must give the results of the default shape
    (points meet other lanes, so the sums round differently: the same trajectories to 1e-9 m in float64, the same
    correspondences and fitness exactly) and the oracle's."""
    from fusion4landslide_amd import synthetic
    d = synthetic.make_patches(60_000, 11, 1.386, seed=23, roughness=0.1)  # 121 patches of ~500 points
    rng = np.random.default_rng(5)
    big = rng.choice(d["P"], 6, replace=False)  # a few patches get three times the targets: the "border patches" of a tile
    tl, to = [], [0]
    for p in range(d["P"]):
        t = d["tgt"][d["tgt_off"][p]:d["tgt_off"][p + 1]]
        if p in big:
            t = np.concatenate([t, t + rng.normal(0, 0.02, t.shape).astype(np.float32), t + rng.normal(0, 0.02, t.shape).astype(np.float32)])
        tl.append(t)
        to.append(to[-1] + len(t))
    tgt, toff = np.concatenate(tl), np.array(to, dtype=np.int64)
    args = (dev(d["src"]), dev(d["src_off"]), dev(tgt), dev(toff))
    kw = dict(max_corr_dist=0.1, max_iter=20, fixed_iters=True, search=search, icp_type=icp_type, return_corr=True)
    monkeypatch.setenv("F4L_ICP_THROUGHPUT", "0")
    ref = eng.piecewise_icp(*args, **kw)
    monkeypatch.setenv("F4L_ICP_THROUGHPUT", "1")
    monkeypatch.setenv("F4L_ICP_PLAN_DEBUG", "1")
    out = eng.piecewise_icp(*args, **kw)
    monkeypatch.delenv("F4L_ICP_THROUGHPUT")
    monkeypatch.delenv("F4L_ICP_PLAN_DEBUG")
    dd = dict(src=d["src"], src_off=d["src_off"], P=d["P"])
    per = _disp_per_patch(dd, out["T"].cpu().numpy(), ref["T"].cpu().numpy())
    assert torch.equal(out["iters"], ref["iters"])
    # (a patch whose blocks moved beyond the correspondence radius is ill-posed: it may settle elsewhere after the first
    #  rounding difference, on any two shapes and against the oracle alike -- DESIGN.md section 4; one such patch here)
    well = (ref["fitness"] > 0.8).cpu().numpy()
    assert well.mean() > 0.6
    if search == "f64":
        assert per[well].max() <= (1e-9 if icp_type == "point2point" else 1e-6), per[well].max()
        wt = torch.from_numpy(well).cuda()
        assert torch.equal(out["fitness"][wt], ref["fitness"][wt])
        assert float((out["rmse"] - ref["rmse"])[wt].abs().max()) < 1e-10
    else:  # float32 mode: ill-posed patches may settle elsewhere after the first rounding difference
        assert np.median(per) <= 1e-5 and (per <= 1e-4).mean() >= 0.85, per
    if search == "f64" and icp_type == "point2point":
        o = O.piecewise_icp(d["src"], d["src_off"], tgt, toff, max_corr_dist=0.1, max_iter=20, fixed_iters=True)
        assert _disp_per_patch(dd, out["T"].cpu().numpy(), o["T"])[well].max() <= 1e-9


def test_icp_vs_open3d_goldens_when_present(eng, golden_dir):
    """Fixtures written by tools/dump_o3d_goldens.py where Open3D 0.19.0 exists; absent in this repository's build
    container (then skipped: the ICP parity is anchored on the CPU oracle)."""
    path = os.path.join(golden_dir, "o3d_icp_golden.npz")
    if not os.path.exists(path):
        pytest.skip("no Open3D fixtures (tools/dump_o3d_goldens.py needs Open3D 0.19.0)")
    g = np.load(path)
    C = int(g["n_cases"])
    src = [g[f"src_{c}"].astype(np.float32) for c in range(C)]
    tgt = [g[f"tgt_{c}"].astype(np.float32) for c in range(C)]
    soff, toff = ragged(None, [len(a) for a in src]), ragged(None, [len(a) for a in tgt])
    T0 = np.stack([g[f"init_{c}"] for c in range(C)])
    for icp_type in ("point2point", "point2plane"):
        out = eng.piecewise_icp(dev(np.concatenate(src)), dev(soff), dev(np.concatenate(tgt)), dev(toff), init_T=dev(T0),
                                max_corr_dist=float(g["threshold"]), max_iter=30, icp_type=icp_type, search="f64", p2plane="open3d")
        T = out["T"].cpu().numpy()
        for c in range(C):
            Tr = g[f"T_{icp_type}_{c}"]
            s = src[c].astype(np.float64)
            assert np.abs((s @ T[c, :3, :3].T + T[c, :3, 3]) - (s @ Tr[:3, :3].T + Tr[:3, 3])).max() <= 1e-6, (icp_type, c)
            assert abs(out["fitness"][c].item() - float(g[f"fitness_{icp_type}_{c}"])) <= 1e-3
            assert abs(out["rmse"][c].item() - float(g[f"rmse_{icp_type}_{c}"])) <= 1e-5
    for tag in ("generalized_icp", "generalized_icp_default"):
        if f"T_{tag}_0" not in g:
            continue
        out = eng.piecewise_icp(dev(np.concatenate(src)), dev(soff), dev(np.concatenate(tgt)), dev(toff), init_T=dev(T0),
                                max_corr_dist=float(g["threshold"]), max_iter=30, icp_type="generalized_icp",
                                gicp_epsilon=float(g[f"epsilon_{tag}"]))
        T = out["T"].cpu().numpy()
        for c in range(C):
            Tr = g[f"T_{tag}_{c}"]
            s = src[c].astype(np.float64)
            assert np.abs((s @ T[c, :3, :3].T + T[c, :3, 3]) - (s @ Tr[:3, :3].T + Tr[:3, 3])).max() <= 1e-6, (tag, c)
            assert abs(out["fitness"][c].item() - float(g[f"fitness_{tag}_{c}"])) <= 1e-3


def test_patch_loop_equals_the_three_launches(eng):
    """f4l_patch_loop = Kabsch init + ICP + displacement rows in one launch; same answers as the separate calls."""
    from fusion4landslide_amd import synthetic
    d = _patches(n=25_000, cells=5, seed=12)
    src, so, tgt, to = dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"])
    P = d["P"]
    eye = torch.eye(4, dtype=torch.float64, device="cuda").repeat(P, 1, 1)
    nn, _ = eng.nn_refine(src, so, tgt, to, eye, torch.full((P,), 0.2, dtype=torch.float64, device="cuda"), return_rows=False)
    cs, ct, coff = synthetic.correspondences_from_nn(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn.cpu().numpy())
    coff[3] = coff[2]  # one patch without correspondences (its rows go to the next one): identity start there
    w = np.random.default_rng(0).uniform(0.1, 1, len(cs)).astype(np.float32)
    for search in ("f64", "f32"):
        T0 = eng.kabsch_transforms(dev(cs), dev(ct), dev(coff), dev(w), 0.2, 1e-6)
        ref = eng.piecewise_icp(src, so, tgt, to, init_T=T0, max_corr_dist=0.1, max_iter=30, search=search, return_corr=True)
        rows_ref = eng.apply_transform(src, so, ref["T"])
        out = eng.patch_loop(src, so, tgt, to, dev(cs), dev(ct), dev(coff), dev(w), 0.2, 1e-6, max_corr_dist=0.1, max_iter=30,
                             search=search, return_corr=True)
        assert np.allclose(T0.cpu().numpy()[2], np.eye(4))
        disp = _disp_per_patch(d, out["T"].cpu().numpy(), ref["T"].cpu().numpy())
        assert disp.max() <= (1e-9 if search == "f64" else 2e-3) and np.median(disp) <= 1e-9
        if search == "f64":
            assert torch.equal(out["iters"], ref["iters"]) and torch.equal(out["corr"], ref["corr"])
        assert torch.equal(out["rows"], eng.apply_transform(src, so, out["T"]))  # the fused rows are apply_transform's
        assert torch.equal(out["rows"][:, :3], src) and (out["rows"] - rows_ref).abs().max().item() <= 2e-3
    Rr, tr = O.kabsch_batched(cs, ct, coff, w, 0.2, 1e-6)
    ref_or = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"],
                             init_T=np.concatenate([np.concatenate([Rr, tr[:, :, None]], 2), np.tile([[[0, 0, 0, 1.0]]], (P, 1, 1))], 1),
                             max_corr_dist=0.1, max_iter=30)
    out = eng.patch_loop(src, so, tgt, to, dev(cs), dev(ct), dev(coff), dev(w), 0.2, 1e-6, max_corr_dist=0.1, max_iter=30, search="f64")
    assert _disp_per_patch(d, out["T"].cpu().numpy(), ref_or["T"]).max() <= 1e-8


def test_apply_transform_and_nn_refine(eng):
    d = _patches(n=12_000, cells=4, seed=6)
    rng = np.random.default_rng(1)
    P = d["P"]
    T = np.tile(np.eye(4), (P, 1, 1))
    for p in range(P):
        T[p, :3, :3] = rot_from_axis_angle(rng.normal(size=3), 0.01)
        T[p, :3, 3] = rng.uniform(-0.05, 0.05, 3)
    rows = eng.apply_transform(dev(d["src"]), dev(d["src_off"]), dev(T)).cpu().numpy()
    inv = eng.apply_transform(dev(d["tgt"]), dev(d["tgt_off"]), dev(T), inverse=True).cpu().numpy()
    thr = np.full(P, 0.08)
    nn, rows6 = eng.nn_refine(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), dev(T), dev(thr))
    nn, rows6 = nn.cpu().numpy(), rows6.cpu().numpy()
    mism = 0
    for p in range(P):
        s0, s1, t0, t1 = d["src_off"][p], d["src_off"][p + 1], d["tgt_off"][p], d["tgt_off"][p + 1]
        s = d["src"][s0:s1].astype(np.float64)
        moved = s @ T[p, :3, :3].T + T[p, :3, 3]
        assert np.array_equal(rows[s0:s1, :3], d["src"][s0:s1])
        assert np.abs(rows[s0:s1, 3:] - moved).max() < 1e-5
        q = d["tgt"][t0:t1].astype(np.float64)
        assert np.abs(inv[t0:t1, :3] - (q - T[p, :3, 3]) @ T[p, :3, :3]).max() < 1e-5
        ref_nn, ref_d2 = O.nn_within(moved, q, thr[p])
        # float32 search vs double KD-tree: only pairs at the threshold / exact NN ties may differ
        mism += int((nn[s0:s1] != ref_nn).sum())
        ok = nn[s0:s1] >= 0
        assert np.array_equal(rows6[s0:s1][ok, 3:], d["tgt"][t0:t1][nn[s0:s1][ok]])
        assert (rows6[s0:s1][~ok, 3:] == 0).all()
    assert mism <= 3e-4 * len(d["src"]) + 2


def test_nn_refine_wave_per_small_patch_is_the_grid_search(eng, monkeypatch):
    """f4l_nn_refine gives patches of up to 128 targets a wave each (all targets measured, no grid; round 5) and the others a
    workgroup over a grid: the same winner -- the minimiser of (float32 d2, index) below the threshold -- and the same rows, on a
    mix of supervoxel-sized patches, patches around the limit (127, 128, 129 targets), empty ones on either side, a zero threshold
    and a patch with duplicated targets; against the KD-tree on top."""
    rng = np.random.default_rng(77)
    sizes_t = [0, 1, 5, 37, 60, 64, 65, 100, 127, 128, 129, 300, 0, 50, 800, 33]
    sizes_s = [12, 0, 70, 64, 1, 130, 65, 40, 200, 128, 90, 310, 0, 75, 500, 64]
    src, tgt, so, to = [], [], [0], [0]
    for ns, nt in zip(sizes_s, sizes_t):
        c = rng.uniform(-40, 40, 3)
        t = (c + rng.uniform(-0.5, 0.5, (nt, 3)) * [1, 1, 0.05]).astype(np.float32)
        if nt == 50:
            t[10:20] = t[0:10]                                       # duplicated targets: the smaller index wins
        s = (c + rng.uniform(-0.5, 0.5, (ns, 3)) * [1, 1, 0.05]).astype(np.float32)
        if nt >= 5 and ns >= 5:
            s[:5] = t[:5]                                            # exact hits (d2 = 0)
        src.append(s); tgt.append(t); so.append(so[-1] + ns); to.append(to[-1] + nt)
    src, tgt = np.concatenate(src), np.concatenate(tgt)
    so, to = np.asarray(so, np.int64), np.asarray(to, np.int64)
    P = len(sizes_t)
    T = np.tile(np.eye(4), (P, 1, 1))
    for p in range(P):
        T[p, :3, :3] = rot_from_axis_angle(rng.normal(size=3), 0.004)
        T[p, :3, 3] = rng.uniform(-0.02, 0.02, 3)
    thr = np.full(P, 0.12)
    thr[3] = 0.0                                                     # no search at all
    args = (dev(src), dev(so), dev(tgt), dev(to), dev(T), dev(thr))
    nn, rows = eng.nn_refine(*args)
    monkeypatch.setenv("F4L_NN_REFINE_NO_SMALL", "1")
    nn_g, rows_g = eng.nn_refine(*args)
    monkeypatch.delenv("F4L_NN_REFINE_NO_SMALL")
    assert torch.equal(nn, nn_g) and torch.equal(rows, rows_g)
    nn_only, none = eng.nn_refine(*args, return_rows=False)
    assert none is None and torch.equal(nn_only, nn)
    # an UNDERSTATED bound (64 for patches of up to 800 targets): every row is still written, with the same answers
    nn_u, rows_u = eng.nn_refine(*args, max_tgt_patch=64)
    assert torch.equal(nn_u, nn) and torch.equal(rows_u, rows)
    nn = nn.cpu().numpy()
    assert (nn[so[3]:so[4]] == -1).all() and (nn[so[0]:so[1]] == -1).all()
    mism = 0
    for p in range(P):
        if sizes_t[p] == 0 or sizes_s[p] == 0 or thr[p] == 0:
            continue
        moved = src[so[p]:so[p + 1]].astype(np.float64) @ T[p, :3, :3].T + T[p, :3, 3]
        ref_nn, _ = O.nn_within(moved, tgt[to[p]:to[p + 1]].astype(np.float64), thr[p])
        mism += int((nn[so[p]:so[p + 1]] != ref_nn).sum())
    assert mism <= 3
    assert (nn[so[13]:so[13] + 5] == np.arange(5)).all()            # the duplicates' first copies


def test_match_lists_are_the_matched_rows_in_order(eng):
    """f4l_match_lists: f4l_nn_refine's answers as the correspondence lists of f4l_patch_loop -- the rows with a match, in row order,
    against their targets, offsets per patch (empty patches, patches without a match, a patch where every row matches)."""
    rng = np.random.default_rng(5)
    ns = [0, 7, 130, 64, 1, 300, 0, 33]
    nt = [4, 0, 90, 64, 2, 500, 0, 10]
    so, to = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64), np.concatenate([[0], np.cumsum(nt)]).astype(np.int64)
    src, tgt = rng.normal(size=(so[-1], 3)).astype(np.float32), rng.normal(size=(to[-1], 3)).astype(np.float32)
    nn = np.full(so[-1], -1, np.int32)
    for p in range(len(ns)):
        if nt[p] == 0:
            continue
        pick = rng.random(ns[p]) < (1.0 if p == 3 else 0.0 if p == 7 else 0.6)
        nn[so[p]:so[p + 1]][pick] = rng.integers(0, nt[p], int(pick.sum()))
    cs, ct, coff = eng.match_lists(dev(src), dev(so), dev(tgt), dev(to), dev(nn))
    keep = nn >= 0
    patch_of = np.repeat(np.arange(len(ns)), ns)
    assert np.array_equal(cs.cpu().numpy(), src[keep]) and np.array_equal(ct.cpu().numpy(), tgt[to[patch_of[keep]] + nn[keep]])
    assert np.array_equal(coff.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(patch_of[keep], minlength=len(ns)))]))
    none = eng.match_lists(dev(src), dev(so), dev(tgt), dev(to), dev(np.full(so[-1], -1, np.int32)))
    assert none[0].shape == (0, 3) and int(none[2].abs().sum()) == 0


# ------------------------------------------------------------------------------- supervoxel partition
def _sv_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "supervoxel_*.npz")))


@pytest.mark.parametrize("name", ["surf_s0_n2000_k15", "surf_s1_n2000_k30", "vol_s2_n2000_k15", "georef_s3_n3000_k30",
                                  "lattice_m24_k9", "surf_s4_n20000_k30", "step_s5_n4000_k12", "slab_s6_n3000_k20"])
def test_knn_normals_supervoxel_vs_golden(eng, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"supervoxel_{name}.npz"))
    xyz, k, res = g["xyz"], int(g["k"]), float(g["resolution"])
    idx, d2 = eng.knn(dev(xyz), k, return_d2=True)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert (idx[:, 0] == np.arange(len(xyz))).mean() > 0.999  # self first (duplicates may tie)
    assert (np.diff(d2, axis=1) >= 0).all()
    if "knn_d2" in g.files:
        assert np.array_equal(d2, g["knn_d2"]), "squared distances must be bit-equal to the reference's"
    ok, row = knn_equal_within_ties(idx, g["knn_idx"], d2)
    assert ok, f"kNN mismatch at query {row}"
    if "lattice" in name:
        return  # exact ties: neighbour order, hence normals/labels, are traversal dependent in the reference
    nrm = eng.normals(dev(xyz), dev(g["knn_idx"])).cpu().numpy()
    assert np.abs(np.sum(nrm * g["normals"], axis=1)).min() >= 1 - 1e-6
    assert np.abs(nrm - g["normals"]).max() <= 1e-9  # same formula, same orientation convention
    labels, K = eng.supervoxel(dev(xyz), k, res)
    labels = labels.cpu().numpy()
    assert K == int(g["n_supervoxels"])
    assert labels.min() == 0 and labels.max() == K - 1
    assert np.array_equal(labels, g["labels"]), f"labels differ at {(labels != g['labels']).mean():.3%} of points"


def test_knn_edge_cases(eng):
    rng = np.random.default_rng(3)
    # clustered + duplicated points, k = 1 and k = 64, n barely above k, collinear cloud
    a = rng.normal(0, 0.01, (300, 3))
    b = rng.normal(0, 1.0, (300, 3)) + 5
    pts = np.concatenate([a, b, a[:50]]).astype(np.float32)  # 50 exact duplicates
    for k in (1, 7, 64):
        idx, d2 = eng.knn(dev(pts), k, return_d2=True)
        ridx, rd2 = O.knn(pts, k)
        assert np.array_equal(d2.cpu().numpy(), rd2)
        ok, row = knn_equal_within_ties(idx.cpu().numpy(), ridx, rd2)
        assert ok, (k, row)
    line = np.zeros((100, 3), np.float32)
    line[:, 0] = np.sort(rng.uniform(0, 1, 100)).astype(np.float32)
    idx, d2 = eng.knn(dev(line), 5, return_d2=True)
    ridx, rd2 = O.knn(line, 5)
    assert np.array_equal(d2.cpu().numpy(), rd2)
    tiny = rng.uniform(0, 1, (9, 3)).astype(np.float32)
    idx, d2 = eng.knn(dev(tiny), 9, return_d2=True)
    assert np.array_equal(d2.cpu().numpy(), O.knn(tiny, 9)[1])
    same = np.ones((40, 3), np.float32)
    idx = eng.knn(dev(same), 4).cpu().numpy()
    assert (np.sort(idx, axis=1) == np.arange(4)).all()  # all distances tie at 0 -> smallest ids
    from fusion4landslide_amd._lib import F4LError
    with pytest.raises(F4LError):
        eng.knn(dev(tiny), 65)


def test_knn_lane_per_query_equals_wave_per_query(eng, monkeypatch):
    """f4l_knn's fast path (one lane per query: scalar-loaded candidates, per-lane d2 histogram, survivors sorted on
    registers; uncertified queries redone one wave each) against the wave-per-query search it replaced: identical indices
    and bit-equal d2 -- on a surface, a volume, a lattice (exact ties everywhere), duplicated points (more ties than the
    survivor list holds: the fallback), a cloud with sparse outliers (queries whose block is too small), and k up to the
    fast path's limit.  The fused normals equal f4l_normals on the same lists bit for bit."""
    rng = np.random.default_rng(3)
    surf = np.c_[rng.uniform(0, 20, (120_000, 2)), np.zeros(120_000)]
    surf[:, 2] = np.sin(surf[:, 0]) * np.cos(0.7 * surf[:, 1]) + rng.normal(0, 0.004, 120_000)
    vol = rng.uniform(0, 3, (60_000, 3))
    gx = np.arange(40, dtype=np.float64) * 0.25
    lattice = np.stack(np.meshgrid(gx, gx, gx[:12], indexing="ij"), -1).reshape(-1, 3)
    dup = np.repeat(rng.uniform(0, 2, (400, 3)), 60, axis=0)  # every point 60 times: 59 exact zero distances
    sparse = np.concatenate([rng.uniform(0, 1, (30_000, 3)) * [5, 5, 0.05], rng.uniform(-40, 40, (300, 3))])
    georef = surf + np.array([2.6e6, 1.2e6, 1500.0])  # Swiss-grid magnitudes: float32 spacing 0.25 m in x
    cases = [("surface", surf, 30), ("volume", vol, 30), ("lattice", lattice, 27), ("duplicates", dup, 30), ("sparse", sparse, 30),
             ("georef", georef, 30), ("k=1", vol, 1), ("k=40", surf[:30_000], 40), ("k=41 (wave path only)", surf[:30_000], 41)]
    for name, pts, k in cases:
        xyz = dev(pts.astype(np.float32))
        monkeypatch.delenv("F4L_KNN_WAVE_PER_QUERY", raising=False)
        idx, nrm, d2 = eng.knn_normals(xyz, k, return_d2=True)
        idx2, d2b = eng.knn(xyz, k, return_d2=True)
        monkeypatch.setenv("F4L_KNN_WAVE_PER_QUERY", "1")
        ridx, rd2 = eng.knn(xyz, k, return_d2=True)
        monkeypatch.delenv("F4L_KNN_WAVE_PER_QUERY")
        assert torch.equal(d2, rd2) and torch.equal(d2b, rd2), name
        assert torch.equal(idx, ridx) and torch.equal(idx2, ridx), name
        assert torch.equal(nrm, eng.normals(xyz, ridx)) or bool((torch.isnan(nrm) == torch.isnan(eng.normals(xyz, ridx))).all()), name
        ok = ~torch.isnan(nrm)
        assert torch.equal(nrm[ok], eng.normals(xyz, ridx)[ok]), name
    # against the CPU oracle on a sample (the wave path is pinned by the golden vectors; this pins the whole chain once more)
    pts = surf[:20_000].astype(np.float32)
    idx, d2 = eng.knn(dev(pts), 30, return_d2=True)
    oi, od = O.knn(pts, 30)
    ok, bad = knn_equal_within_ties(idx.cpu().numpy(), oi, d2.cpu().numpy(), od)
    assert ok, bad


def test_labels_to_csr_and_gather(eng):
    rng = np.random.default_rng(4)
    K, n = 37, 5000
    labels = rng.integers(0, K, n).astype(np.int32)
    labels[labels == 5] = 6  # an empty supervoxel
    order, off = eng.labels_to_csr(dev(labels), K)
    order, off = order.cpu().numpy(), off.cpu().numpy()
    assert off[0] == 0 and off[-1] == n and np.array_equal(np.diff(off), np.bincount(labels, minlength=K))
    assert np.array_equal(order, np.argsort(labels, kind="stable"))
    pts = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    assert np.array_equal(eng.gather_points(dev(pts), dev(order.astype(np.int32))).cpu().numpy(), pts[order])
    # empty supervoxels at both ends and in a row, a single label, a single point
    for lab, k in ((np.where((labels < 3) | (labels > 30), 17, labels), K), (np.zeros(100, np.int32), 1), (np.full(7, 4, np.int32), 9),
                   (np.array([2], np.int32), 5)):
        o, f = eng.labels_to_csr(dev(lab.astype(np.int32)), k)
        assert np.array_equal(np.diff(f.cpu().numpy()), np.bincount(lab, minlength=k)) and int(f[0]) == 0
        assert np.array_equal(o.cpu().numpy(), np.argsort(lab, kind="stable"))
    # labels outside [0, K) -- an "unlabelled" -1, labels beyond the caller's count -- belong to no patch: their points follow the
    # last patch in `order`, the offsets are those of the labelled points alone (they used to land by their low bits)
    bad = labels.copy()
    bad[rng.choice(n, 300, replace=False)] = -1
    bad[rng.choice(n, 200, replace=False)] = K + rng.integers(0, 1000, 200).astype(np.int32)
    bad[-1] = -7
    o, f = eng.labels_to_csr(dev(bad), K)
    o, f = o.cpu().numpy(), f.cpu().numpy()
    ok = (bad >= 0) & (bad < K)
    assert f[0] == 0 and f[-1] == ok.sum() and np.array_equal(np.diff(f), np.bincount(bad[ok], minlength=K))
    key = np.where(ok, bad, K)
    assert np.array_equal(o, np.argsort(key, kind="stable"))


# --------------------------------------------------------------- size-independent properties, full size
def test_full_size_properties_1M(eng):
    """BASELINE.json config 2 (1 M points, 45 x 45 patches, 20 fixed iterations): the oracle is too slow to run here
    in full, so check properties that do not need it."""
    from fusion4landslide_amd import synthetic
    d = synthetic.make_patches(1_000_000, 45, 1.386, seed=0)
    src, so, tgt, to = dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"])
    out = eng.piecewise_icp(src, so, tgt, to, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search="f32",
                            max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
    T = out["T"].cpu().numpy()
    assert (out["iters"].cpu().numpy() == 20).all()
    R = T[:, :3, :3]
    assert np.abs(R @ np.transpose(R, (0, 2, 1)) - np.eye(3)).max() < 1e-9 and np.allclose(np.linalg.det(R), 1.0)
    fit, rmse = out["fitness"].cpu().numpy(), out["rmse"].cpu().numpy()
    assert ((fit >= 0) & (fit <= 1)).all() and (rmse < 0.1).all()
    # idempotence: restarting from the converged transforms changes nothing beyond the tolerance
    again = eng.piecewise_icp(src, so, tgt, to, init_T=out["T"], max_corr_dist=0.1, max_iter=30,
                              max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
    conv = out["iters"].cpu().numpy() > 0
    Ta = again["T"].cpu().numpy()
    moved = np.abs(Ta[:, :3, 3] - T[:, :3, 3]).max(axis=1)
    assert np.median(moved) < 1e-4
    # a bounded sample of patches against the oracle: the float64 search (parity mode) must reproduce it, the float32
    # fast path is held to its stated tolerance (see test_icp_point2point_vs_oracle)
    out64 = eng.piecewise_icp(src, so, tgt, to, max_corr_dist=0.1, max_iter=20, fixed_iters=True, search="f64",
                              max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
    T64 = out64["T"].cpu().numpy()
    pick = np.linspace(0, d["P"] - 1, 24).astype(int)
    dev32, dev64 = [], []
    for p in pick:
        s0, s1, t0, t1 = d["src_off"][p], d["src_off"][p + 1], d["tgt_off"][p], d["tgt_off"][p + 1]
        one = O.icp(d["src"][s0:s1], d["tgt"][t0:t1], max_corr_dist=0.1, max_iter=20, fixed_iters=True)
        s = d["src"][s0:s1].astype(np.float64)
        b = s @ one["est_transform"][:3, :3].T + one["est_transform"][:3, 3]
        dev32.append(np.abs(s @ T[p, :3, :3].T + T[p, :3, 3] - b).max())
        dev64.append(np.abs(s @ T64[p, :3, :3].T + T64[p, :3, 3] - b).max())
        assert abs(out64["fitness"].cpu().numpy()[p] - one["fitness"]) < 1e-12, p
    dev32, dev64 = np.array(dev32), np.array(dev64)
    assert dev64.max() <= 1e-9, dev64.max()
    # (ill-posed patches -- displaced beyond the radius, fitness well below 1 -- can land in another local solution
    #  altogether in float32: no cap on the worst case)
    assert np.median(dev32) <= 1e-5 and (dev32 <= 1e-4).mean() >= 0.85, dev32
    rows = eng.apply_transform(src, so, out["T"])
    assert rows.shape == (1_000_000, 6) and torch.equal(rows[:, :3], src)
    # kNN at full size: sortedness, self first, checksum against a sampled oracle
    idx, d2 = eng.knn(src, 30, return_d2=True)
    assert bool((d2[:, 1:] >= d2[:, :-1]).all()) and bool((d2[:, 0] == 0).all())
    assert float((idx[:, 0] == torch.arange(1_000_000, device="cuda")).float().mean()) > 0.9999
    assert conv.any()


def test_icp_launch_is_capturable_into_a_hip_graph(eng):
    """The C ABI only enqueues on the caller's stream (the size classes fork to helper streams and join again through
    events, temporaries come from the stream-ordered allocator): a launch can be captured once and replayed.  Checked for
    the single-launch path and for an uneven batch that takes the size-class path."""
    rng = np.random.default_rng(41)
    cases = [_patches(n=20_000, cells=5, seed=19)]
    sizes = rng.integers(10, 300, 700)
    src_l, tgt_l = [], []
    for m in sizes:
        side = max(0.2, np.sqrt(m / 400.0))
        xy = rng.uniform(0, side, (int(m), 2))
        t = np.c_[xy, 0.3 * np.sin(1.7 * xy[:, 0]) * np.cos(2.3 * xy[:, 1])]
        tgt_l.append(t)
        src_l.append(t + rng.uniform(-0.02, 0.02, 3) + rng.normal(0, 0.002, t.shape))
    off = ragged(rng, [int(m) for m in sizes])
    cases.append(dict(src=np.concatenate(src_l).astype(np.float32), tgt=np.concatenate(tgt_l).astype(np.float32),
                      src_off=off, tgt_off=off.copy(), max_src=int(sizes.max()), max_tgt=int(sizes.max())))
    for d in cases:
        args = (dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]))
        kw = dict(max_corr_dist=0.1, max_iter=20, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
        ref = eng.piecewise_icp(*args, **kw)
        torch.cuda.synchronize()
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            eng.piecewise_icp(*args, **kw)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = eng.piecewise_icp(*args, **kw)
        for _ in range(3):
            out["T"].zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out["T"], ref["T"]) and torch.equal(out["iters"], ref["iters"])


def test_patch_loop_over_several_tiles_in_one_launch(eng):
    """engine.patch_loop_tiles: the loop body of several tiles (the reference's unit of work, main_fusion.py:134) concatenated into
    one launch gives every tile the results of its own launch -- to rounding when the merged batch crosses into the throughput
    launch shape (sums in another order), bit for bit otherwise -- and its own rows."""
    from fusion4landslide_amd import synthetic
    tiles = []
    for seed in (0, 1, 2):
        d = synthetic.make_patches(30_000, 7, 1.386, seed=seed, roughness=0.05)
        t = {k: dev(d[k]) for k in ("src", "src_off", "tgt", "tgt_off")}
        eye = torch.eye(4, dtype=torch.float64, device="cuda").repeat(d["P"], 1, 1)
        nn, _ = eng.nn_refine(t["src"], t["src_off"], t["tgt"], t["tgt_off"], eye, torch.full((d["P"],), 0.2, dtype=torch.float64, device="cuda"),
                              return_rows=False)
        cs, ct, coff = synthetic.correspondences_from_nn(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn.cpu().numpy())
        t.update(corr_src=dev(cs), corr_ref=dev(ct), corr_off=dev(coff), max_src=d["max_src"], max_tgt=d["max_tgt"])
        tiles.append(t)
    kw = dict(max_corr_dist=0.1, max_iter=20, fixed_iters=True)
    merged = eng.patch_loop_tiles(tiles, **kw)
    assert len(merged) == 3
    for t, m in zip(tiles, merged):
        one = eng.patch_loop(t["src"], t["src_off"], t["tgt"], t["tgt_off"], t["corr_src"], t["corr_ref"], t["corr_off"], **kw)
        assert m["T"].shape == one["T"].shape and m["rows"].shape == one["rows"].shape == (t["src"].shape[0], 6)
        assert torch.equal(m["iters"], one["iters"]) and torch.equal(m["rows"][:, :3], t["src"])
        assert float((m["T"] - one["T"]).abs().max()) < 1e-9 and float((m["rows"] - one["rows"]).abs().max()) < 1e-5


@pytest.mark.parametrize("search", ["f64", "f32"])
def test_icp_cell_shape_changes_no_answer(eng, search, monkeypatch):
    """The grid of a target patch is a search structure only: the answer of every search is the minimiser of (d2, original
    index) whatever the cells look like.  So the column grids of flat patches and the finer cells along x (patch_grid.h "Cell
    shape", round 4) must give what the cubic cells of rounds 1-3 give (F4L_ICP_DEBUG bits 512 + 1024) -- on terrain patches
    (columns), on a volume (layers), on a dense patch (cells finer than the radius: the wide stencils):
      * with every point searched in every pass (bit 4: no certificates) BIT FOR BIT: the sums then run in point order;
      * with certificates the scanned candidates, hence the runner-up distances, hence WHICH points are searched and in which
        lane their pair is summed, depend on the cells: the same trajectories to rounding (float64: 1e-9 m, equal iteration
        counts, fitness and final correspondences).
    (f4l_nn_refine shares the grid; its tests hold it to the oracle's KD-tree.)"""
    from fusion4landslide_amd import synthetic
    d = synthetic.make_patches(40_000, 6, 1.386, seed=5, roughness=0.1)
    rng = np.random.default_rng(9)
    vol_t = rng.uniform(0, 0.6, (900, 3)).astype(np.float32)            # a volume: layered cells
    vol_s = (vol_t[:700] + rng.normal(0, 0.004, (700, 3))).astype(np.float32)
    xy = rng.uniform(0, 0.25, (2500, 2))                                  # dense against the radius: a subdivided grid
    den_t = np.c_[xy, 0.02 * np.sin(20 * xy[:, 0])].astype(np.float32)
    den_s = (den_t[:2000] + rng.normal(0, 0.002, (2000, 3))).astype(np.float32)
    src = np.concatenate([d["src"], vol_s, den_s])
    tgt = np.concatenate([d["tgt"], vol_t, den_t])
    soff = np.r_[d["src_off"], d["src_off"][-1] + len(vol_s), d["src_off"][-1] + len(vol_s) + len(den_s)].astype(np.int64)
    toff = np.r_[d["tgt_off"], d["tgt_off"][-1] + len(vol_t), d["tgt_off"][-1] + len(vol_t) + len(den_t)].astype(np.int64)
    args = (dev(src), dev(soff), dev(tgt), dev(toff))
    kw = dict(max_corr_dist=0.1, max_iter=20, fixed_iters=True, search=search, return_corr=True)
    out = eng.piecewise_icp(*args, **kw)
    assert float((out["fitness"] > 0.3).double().mean()) > 0.4 and float(out["fitness"][-2:].min()) > 0.9  # (blocks moved out of reach: 0)
    monkeypatch.setenv("F4L_ICP_DEBUG", "4")
    plain = eng.piecewise_icp(*args, **kw)
    dd = dict(src=src, src_off=soff, P=len(soff) - 1)
    for bits in (512, 1024, 1536):
        monkeypatch.setenv("F4L_ICP_DEBUG", str(4 + bits))
        ref = eng.piecewise_icp(*args, **kw)
        for key in ("T", "fitness", "rmse", "iters", "corr"):
            assert torch.equal(plain[key], ref[key]), (bits, key)
        monkeypatch.setenv("F4L_ICP_DEBUG", str(bits))
        ref = eng.piecewise_icp(*args, **kw)
        if search == "f64":
            assert _disp_per_patch(dd, out["T"].cpu().numpy(), ref["T"].cpu().numpy()).max() <= 1e-9, bits
            for key in ("fitness", "iters", "corr"):
                assert torch.equal(out[key], ref[key]), (bits, key)
        else:
            per = _disp_per_patch(dd, out["T"].cpu().numpy(), ref["T"].cpu().numpy())
            assert np.median(per) <= 1e-5 and (per <= 1e-4).mean() >= 0.85, (bits, per)
    monkeypatch.delenv("F4L_ICP_DEBUG")
