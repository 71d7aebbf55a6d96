"""CPU suite, part 1: the oracle against the committed golden vectors (and, when it was built in a
container holding /root/reference, against the reference's own compiled templates)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests._util import knn_equal_within_ties, rotation_angle


def _sv_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "supervoxel_*.npz")))


def test_golden_files_present(golden_dir):
    assert len(_sv_cases(golden_dir)) >= 5
    assert os.path.exists(os.path.join(golden_dir, "kabsch_golden.npz"))


@pytest.mark.parametrize("name", ["surf_s0_n2000_k15", "surf_s1_n2000_k30", "vol_s2_n2000_k15",
                                  "georef_s3_n3000_k30", "lattice_m24_k9", "surf_s4_n20000_k30", "step_s5_n4000_k12", "slab_s6_n3000_k20"])
def test_supervoxel_oracle_vs_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"supervoxel_{name}.npz"))
    xyz, k, res = g["xyz"], int(g["k"]), float(g["resolution"])
    out = O.supervoxel(xyz, k, res)
    # kNN: bit-exact d2 (same double arithmetic), indices equal modulo exact-tie order
    if "knn_d2" in g.files:
        assert np.array_equal(out["knn_d2"], g["knn_d2"])
    ok, row = knn_equal_within_ties(out["knn_idx"], g["knn_idx"], out["knn_d2"])
    assert ok, f"kNN mismatch at query {row}"
    assert O.grid_cell_count(xyz, res) == int(g["n_grid_cells"])
    if "lattice" in name:
        # exact ties make neighbour order (and therefore everything downstream) traversal dependent:
        # feed the reference's own neighbour lists to pin normals + segmentation
        nrm = O.normals_from_knn(xyz, g["knn_idx"])
        finite = np.isfinite(g["normals"]).all(1)
        assert np.array_equal(np.isfinite(nrm).all(1), finite)
        assert np.allclose(nrm[finite], g["normals"][finite], atol=1e-12)
        labels, nsv, _ = O.supervoxel_segment(xyz, np.nan_to_num(g["normals"]), g["knn_idx"], res)
        ref_labels = g["labels"]
        if finite.all():
            assert nsv == int(g["n_supervoxels"])
            assert np.array_equal(labels, ref_labels)
        return
    dots = np.abs(np.sum(out["normals"] * g["normals"], axis=1))
    assert dots.min() >= 1.0 - 1e-12
    assert out["n_supervoxels"] == int(g["n_supervoxels"])
    assert np.array_equal(out["labels"], g["labels"]), "labels differ from the reference's"


def test_supervoxel_oracle_vs_large_reference_fixture(golden_dir):
    """The reference-held partition of a 300 k-point cloud (tests/golden/sv_large_ref.npz: labels, K, grid cells, lambda0 and
    checksums produced by the reference's own templates, tools/make_golden_supervoxel.py --large; the cloud comes back from its
    seed): 15 x the largest small fixture.  The C oracle reproduces every label, and its neighbour lists and squared distances
    checksum to the reference's bits."""
    from _util import LARGE_CASE, bits_checksum, large_surface_cloud
    g = np.load(os.path.join(golden_dir, "sv_large_ref.npz"))
    c = LARGE_CASE
    xyz = large_surface_cloud(c["seed"], c["n"], c["extent"])
    assert bits_checksum(xyz) == int(g["xyz_checksum"]), "the generator no longer reproduces the fixture's cloud"
    O.set_threads(0)  # (kNN and normals of 300 k points: all cores; the segmentation itself is sequential)
    try:
        out = O.supervoxel(xyz, c["k"], c["resolution"])
    finally:
        O.set_threads(1)
    assert bits_checksum(out["knn_idx"]) == int(g["knn_idx_checksum"])
    assert bits_checksum(out["knn_d2"]) == int(g["knn_d2_checksum"])
    assert np.allclose(np.abs(out["normals"]).sum(axis=0), g["normals_abs_sum"], rtol=1e-12)
    assert O.grid_cell_count(xyz, c["resolution"]) == int(g["n_grid_cells"]) == int(g["n_supervoxels"]) == out["n_supervoxels"]
    assert np.array_equal(out["labels"], g["labels"])


def test_supervoxel_oracle_vs_live_reference():
    if not O.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    rng = np.random.default_rng(77)
    xyz = rng.uniform(0, 1, (1500, 3)).astype(np.float32)
    xyz[:, 2] *= 0.05
    a, b = O.supervoxel(xyz, 12, 0.2), O.ref_supervoxel(xyz, 12, 0.2)
    assert np.array_equal(a["knn_idx"], b["knn_idx"])
    assert np.array_equal(a["knn_d2"], b["knn_d2"])
    assert np.allclose(a["normals"], b["normals"], atol=1e-13)
    assert a["n_supervoxels"] == b["n_supervoxels"] == b["n_grid_cells"]
    assert np.array_equal(a["labels"], b["labels"])
    # single neighbourhood, including the degenerate all-identical case (NaN in both)
    nb = rng.normal(size=(20, 3))
    assert np.allclose(O.pca_normal(nb), O.ref_pca_normal(nb), atol=1e-14)
    same = np.ones((5, 3))
    assert np.isnan(O.pca_normal(same)).all() and np.isnan(O.ref_pca_normal(same)).all()


def test_supervoxel_label_invariants():
    rng = np.random.default_rng(5)
    xyz = rng.uniform(0, 2, (4000, 3)).astype(np.float32)
    xyz[:, 2] = 0.2 * np.sin(3 * xyz[:, 0])
    out = O.supervoxel(xyz, 20, 0.4)
    labels, K = out["labels"], out["n_supervoxels"]
    assert K == O.grid_cell_count(xyz, 0.4)
    assert labels.min() == 0 and labels.max() == K - 1
    assert np.unique(labels).shape[0] == K


def _kabsch_cases(g):
    names = sorted({k.rsplit("_", 1)[0] for k in g.files if k.startswith("c") and k.endswith("_R")})
    return names



def _rank_deficient(src, tgt, w, thr, eps):
    if src.ndim != 2:
        return False
    s, t = src.astype(np.float64), tgt.astype(np.float64)
    ww = np.ones(len(s)) if w is None else np.where(w < thr, 0.0, w.astype(np.float64))
    ww = ww / (ww.sum() + eps)
    cs, ct = (s * ww[:, None]).sum(0), (t * ww[:, None]).sum(0)
    sv = np.linalg.svd((s - cs).T @ (ww[:, None] * (t - ct)), compute_uv=False)
    return sv[1] < 1e-7 * sv[0]


def _check_deficient(src, tgt, w, thr, R, t, R_ref, t_ref, tol):
    assert np.allclose(R @ R.T, np.eye(3), atol=tol)
    keep = np.ones(len(src), bool) if w is None else (w >= thr)
    a = src[keep].astype(np.float64) @ np.asarray(R, np.float64).T + t
    b = src[keep].astype(np.float64) @ np.asarray(R_ref, np.float64).T + t_ref
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_weighted_procrustes_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "kabsch_golden.npz"))
    names = _kabsch_cases(g)
    assert len(names) >= 30
    for nm in names:
        src, tgt = g[nm + "_src"], g[nm + "_tgt"]
        w = g[nm + "_w"] if nm + "_w" in g.files else None
        eps, thr = float(g[nm + "_eps"]), float(g[nm + "_thr"])
        R_ref, t_ref = g[nm + "_R"], g[nm + "_t"]
        fp32 = src.dtype == np.float32
        # numpy restatement in the fixture's own dtype
        R, t = O.weighted_procrustes(src, tgt, w, thr, eps, dtype=src.dtype.type)
        scale = max(1.0, float(np.abs(src).max()))
        rtol = 5e-5 if fp32 else 1e-9
        ttol = (2e-4 if fp32 else 1e-9) * scale
        if "georef" in nm and fp32:
            rtol, ttol = 2e-2, 2.0  # fp32 at km-scale coordinates is ill-conditioned in the reference itself
        if _rank_deficient(src, tgt, w, thr, eps):
            # <3 effective correspondences: H has rank 1, the rotation about the surviving direction is
            # arbitrary (whatever LAPACK returns in the reference).  Pin what IS determined: R is
            # orthonormal and maps the weighted points where the reference's R does.
            _check_deficient(src, tgt, w, thr, R, t, R_ref, t_ref, 1e-3 if fp32 else 1e-8)
            Rc, tc = O.weighted_procrustes_c(src, tgt, w, thr, eps)
            _check_deficient(src, tgt, w, thr, Rc, tc, R_ref, t_ref, 1e-3 if fp32 else 1e-8)
            continue
        assert np.abs(R - R_ref).max() <= rtol, nm
        assert np.abs(t - t_ref).max() <= ttol, nm
        # C restatement (double) on each batch element; fp32 fixtures are compared loosely
        S = src if src.ndim == 3 else src[None]
        T = tgt if tgt.ndim == 3 else tgt[None]
        W = None if w is None else (w if w.ndim == 2 else w[None])
        Rr = R_ref if R_ref.ndim == 3 else R_ref[None]
        tr = t_ref if t_ref.ndim == 2 else t_ref[None]
        for b in range(S.shape[0]):
            Rc, tc = O.weighted_procrustes_c(S[b], T[b], None if W is None else W[b], thr, eps)
            assert np.abs(Rc - Rr[b]).max() <= rtol, nm
            assert np.abs(tc - tr[b]).max() <= ttol, nm


def test_kabsch2_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "kabsch_golden.npz"))
    for j in range(2):
        R, t = O.kabsch_transformation_estimation(g[f"k2_{j}_x1"], g[f"k2_{j}_x2"], g[f"k2_{j}_w"])
        assert np.abs(R - g[f"k2_{j}_R"]).max() < 1e-9
        assert np.abs(t - g[f"k2_{j}_t"]).max() < 1e-9
        R, t = O.kabsch_transformation_estimation(g[f"k2_{j}_x1"], g[f"k2_{j}_x2"], None)
        assert np.abs(R - g[f"k2_{j}_R_now"]).max() < 1e-9
        assert np.abs(t - g[f"k2_{j}_t_now"]).max() < 1e-9


def test_svd3_properties():
    rng = np.random.default_rng(3)
    mats = [rng.normal(size=(3, 3)) for _ in range(50)]
    mats += [np.outer(rng.normal(size=3), rng.normal(size=3)), np.zeros((3, 3)), np.diag([3.0, 3.0, 1.0]),
             np.diag([1.0, 1.0, 0.0]) @ rng.normal(size=(3, 3))]
    for A in mats:
        U, S, V = O.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert S[0] >= S[1] >= S[2] >= 0
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-12)


def test_refine_prune_mask_hand_case():
    # scripts/weighted_svd.py:143-147: rows with residual >= 1 m are dropped
    rng = np.random.default_rng(9)
    src = rng.uniform(-1, 1, (50, 3))
    tgt = src + np.array([0.1, 0.0, 0.0])
    tgt[7] += np.array([0.0, 3.0, 0.0])  # one gross outlier
    pruned, T, keep = O.refine_local_rigid_correspondences(np.c_[src, tgt])
    assert not keep[7] and keep.sum() == 49 and pruned.shape == (49, 6)
    assert rotation_angle(T[:3, :3], np.eye(3)) < 0.5
