"""GPU tests of the host-side mirrors of the reference interfaces (same names / arguments / error behaviour)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402
from tests._util import rot_from_axis_angle  # noqa: E402


def test_weighted_procrustes_mirror_vs_golden(golden_dir):
    from fusion4landslide_amd.scripts.weighted_svd import (refine_local_rigid_correspondences, weighted_procrustes,
                                                           weighted_svd)
    g = np.load(os.path.join(golden_dir, "kabsch_golden.npz"))
    names = sorted({k.rsplit("_", 1)[0] for k in g.files if k.startswith("c") and k.endswith("_R")})
    n_checked = 0
    for nm in names:
        src, tgt = g[nm + "_src"], g[nm + "_tgt"]
        w = g[nm + "_w"] if nm + "_w" in g.files else None
        if w is not None and src.ndim == 2 and int((w >= float(g[nm + "_thr"])).sum()) < 3:
            continue
        if "georef" in nm and src.dtype == np.float32:
            continue
        R, t = weighted_procrustes(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(),
                                   None if w is None else torch.from_numpy(w).cuda(), float(g[nm + "_thr"]),
                                   float(g[nm + "_eps"]), return_transform=False)
        assert R.dtype == torch.from_numpy(src).dtype and tuple(R.shape) == g[nm + "_R"].shape
        tol = 5e-5 if src.dtype == np.float32 else 1e-9
        scale = max(1.0, float(np.abs(src).max()))
        assert np.abs(R.cpu().numpy() - g[nm + "_R"]).max() <= tol, nm
        assert np.abs(t.cpu().numpy() - g[nm + "_t"]).max() <= (2e-4 if src.dtype == np.float32 else 1e-9) * scale, nm
        n_checked += 1
    assert n_checked >= 30
    # transform form, older variant, pruning
    rng = np.random.default_rng(0)
    src = rng.uniform(-1, 1, (60, 3)).astype(np.float32)
    R0 = rot_from_axis_angle([1, 2, 3], 0.3)
    tgt = (src @ R0.T + [0.1, -0.2, 0.3]).astype(np.float32)
    T = weighted_procrustes(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda())
    assert tuple(T.shape) == (4, 4) and np.abs(T.cpu().numpy()[:3, :3] - R0).max() < 1e-5
    T2 = weighted_svd(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), weights=torch.ones(60, 1).cuda())
    assert np.abs(T2.cpu().numpy() - T.cpu().numpy()).max() < 1e-5
    tgt[5] += 3.0
    corr = torch.from_numpy(np.c_[src, tgt]).cuda()
    pruned, T3 = refine_local_rigid_correspondences(corr)
    ref_pruned, ref_T, keep = O.refine_local_rigid_correspondences(np.c_[src, tgt])
    assert pruned.shape[0] == keep.sum() == 59 and np.abs(T3.cpu().numpy() - ref_T).max() < 1e-4
    with pytest.raises(NotImplementedError):
        refine_local_rigid_correspondences(corr, refine_type='RANSAC')


def test_icp_registration_mirror():
    from fusion4landslide_amd import synthetic
    from fusion4landslide_amd.utils.o3d_tools import icp_registration, tensor2pcd
    d = synthetic.make_patches(3000, 2, 1.386, seed=3, roughness=0.15)
    s = d["src"][d["src_off"][0]:d["src_off"][1]]
    t = d["tgt"][d["tgt_off"][0]:d["tgt_off"][1]]
    src_pcd, tgt_pcd = tensor2pcd(torch.from_numpy(s)), tensor2pcd(torch.from_numpy(t))
    for icp_type in ("point2point", "point2plane"):
        res = icp_registration(src_pcd, tgt_pcd, torch.eye(4), threshold=0.1, icp_type=icp_type)
        ref = O.icp(s, t, np.eye(4), 0.1, 30, icp_type=icp_type)
        assert set(res) == {"fitness", "inlier_rmse", "correspondence_set", "est_transform", "src_corr_pts", "tgt_corr_pts"}
        assert res["est_transform"].dtype == np.float64 and res["est_transform"].shape == (4, 4)
        assert np.abs(res["est_transform"] - ref["est_transform"]).max() < 1e-6
        assert abs(res["fitness"] - ref["fitness"]) < 1e-12 and abs(res["inlier_rmse"] - ref["inlier_rmse"]) < 1e-8
        assert np.array_equal(res["correspondence_set"], ref["correspondence_set"])
        assert res["src_corr_pts"].shape == res["tgt_corr_pts"].shape == (len(ref["correspondence_set"]), 3)
    assert tgt_pcd.has_normals() and src_pcd.has_normals()  # the reference mutates its inputs too
    # icp_type 'generalized_icp' (:40-41, 51-56): the reference's estimator is built with `False` where Open3D expects epsilon
    res = icp_registration(tensor2pcd(torch.from_numpy(s)), tensor2pcd(torch.from_numpy(t)), np.eye(4), threshold=0.1,
                           icp_type="generalized_icp")
    ref = O.gicp(s, t, np.eye(4), 0.1, 30, epsilon=0.0)
    assert np.abs(res["est_transform"] - ref["est_transform"]).max() < 1e-6 and res["fitness"] == ref["fitness"]
    assert np.array_equal(res["correspondence_set"], ref["correspondence_set"])
    res = icp_registration(tensor2pcd(torch.from_numpy(s)), tensor2pcd(torch.from_numpy(t)), np.eye(4), threshold=0.1,
                           icp_type="generalized_icp", gicp_epsilon=1e-3)
    ref = O.gicp(s, t, np.eye(4), 0.1, 30, epsilon=1e-3)
    assert np.abs(res["est_transform"] - ref["est_transform"]).max() < 1e-7 and res["fitness"] == ref["fitness"]
    with pytest.raises(ValueError):
        icp_registration(src_pcd, tgt_pcd, np.eye(4), icp_type="point2line")
    # float64 clouds at georeferenced magnitudes (Open3D's clouds are double): float32 spacing there is 0.25 m, above the
    # 0.1 m correspondence distance -- the mirror moves both clouds to a local origin in double before its float32 cast
    off = np.array([2647123.4, 1177456.7, 1500.2])
    s64, t64 = s.astype(np.float64) + off, t.astype(np.float64) + off
    init = np.eye(4)
    init[:3, 3] = [0.01, -0.02, 0.005]
    res = icp_registration(tensor2pcd(torch.from_numpy(s64)), tensor2pcd(torch.from_numpy(t64)), init, threshold=0.1)
    ref = O.icp(s64, t64, init, 0.1, 30)
    moved = s64 @ res["est_transform"][:3, :3].T + res["est_transform"][:3, 3]
    want = s64 @ ref["est_transform"][:3, :3].T + ref["est_transform"][:3, 3]
    assert np.abs(moved - want).max() < 2e-4  # float32 of the local coordinates (1e-7 x ~3 m), not of the georeferenced ones
    assert abs(res["fitness"] - ref["fitness"]) < 5e-3 and res["fitness"] > 0.5


def test_compute_supervoxel_shim_and_partition_file(golden_dir, tmp_path):
    from fusion4landslide_amd.cpp_core.supervoxel_segmentation.build import supervoxel
    from fusion4landslide_amd.utils.ply import read_ply, write_ply
    g = np.load(os.path.join(golden_dir, "supervoxel_surf_s0_n2000_k15.npz"))
    ply = str(tmp_path / "cloud.ply")
    write_ply(ply, g["xyz"])
    xyz_back, _ = read_ply(ply)
    assert np.array_equal(xyz_back.astype(np.float32), g["xyz"])
    out_txt = str(tmp_path / "partition.txt")
    ret = supervoxel.computeSupervoxel(ply, int(g["k"]), float(g["resolution"]), out_txt)
    labels = np.asarray(ret)
    assert len(ret) == 2000 and np.array_equal(labels, g["labels"])
    assert np.asarray(ret).reshape(-1, 1).shape == (2000, 1)  # src/rgb_guided.py:888 usage
    table = np.loadtxt(out_txt)  # what load_partition does (src/coarse_to_fine_matching_base.py:1257)
    assert table.shape == (2000, 7)
    assert np.array_equal(table[:, 6].astype(np.int64), g["labels"])
    assert np.allclose(table[:, :3], g["xyz"].astype(np.float64), rtol=1e-11, atol=0)
    # one colour per supervoxel from a default-seeded mt19937 (supervoxel.cpp:50-54)
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)
    raw = bg.random_raw(int(g["n_supervoxels"]))
    rgb = np.stack([(raw >> 16) & 0xff, (raw >> 8) & 0xff, raw & 0xff], axis=1)
    assert np.array_equal(table[:, 3:6].astype(np.int64), rgb[g["labels"]])
    ret2 = supervoxel.computeSupervoxel(ply, int(g["k"]), float(g["resolution"]), "None")
    assert np.array_equal(np.asarray(ret2), labels)
    with pytest.raises(ValueError):
        supervoxel.computeSupervoxel(ply, 2000, 0.1)
    # the all-device mode behind the same call: same K, a valid partition file, not the reference's labels
    supervoxel.SEGMENTATION = "parallel"
    try:
        ret3 = np.asarray(supervoxel.computeSupervoxel(ply, int(g["k"]), float(g["resolution"]), out_txt))
    finally:
        supervoxel.SEGMENTATION = "identical"
    assert ret3.shape == (2000,) and ret3.min() == 0 and ret3.max() == int(g["n_supervoxels"]) - 1
    assert len(np.unique(ret3)) == int(g["n_supervoxels"])
    assert np.array_equal(np.loadtxt(out_txt)[:, 6].astype(np.int64), ret3)


def test_piecewise_icp_entry_writes_reference_files(tmp_path):
    from fusion4landslide_amd import synthetic
    from fusion4landslide_amd.src.piecewise_icp import Piecewise_ICP
    from fusion4landslide_amd.utils.common import AttrDict, get_logger
    from fusion4landslide_amd.utils.ply import write_ply
    c = synthetic.two_epoch_cloud(50_000, 8, 1.386, seed=0)
    write_ply(str(tmp_path / "source_tile_0_overlap.ply"), c["src"])
    write_ply(str(tmp_path / "target_tile_0_overlap.ply"), c["tgt"])
    for eng_name in ("reference_octree", "patch_icp"):
        out_root = tmp_path / eng_name
        cfg = AttrDict(src_tile_overlap_path=str(tmp_path / "source_tile_0_overlap.ply"),
                       tgt_tile_overlap_path=str(tmp_path / "target_tile_0_overlap.ply"), smax=1.4, number_points_min=10,
                       threshold=0.1, output_root=str(out_root), tile_id="0", dataset="brienz_tls",
                       logging=get_logger(), engine=eng_name)
        assert Piecewise_ICP(cfg) is None
        dvfs = np.loadtxt(out_root / "results" / "piecewise_icp_dvfs_of_tile_0.txt")
        dvfms = np.loadtxt(out_root / "results" / "piecewise_icp_dvfms_of_tile_0.txt")
        vis = np.loadtxt(out_root / "results" / "piecewise_dvfms_visualize_of_tile_0.txt")
        assert dvfs.shape[1] == 6 and dvfms.shape == (dvfs.shape[0], 4) and vis.shape == dvfms.shape
        assert dvfs.shape[0] > 40_000
        assert np.allclose(dvfms[:, 3], np.linalg.norm(dvfs[:, :3] - dvfs[:, 3:], axis=1))
        assert vis[0, 3] == 0 and vis[1, 3] == 5
        if eng_name == "reference_octree":
            # stable cells keep their points, unstable ones move rigidly by a centroid difference
            still = np.all(dvfs[:, :3] == dvfs[:, 3:], axis=1)
            assert 0.3 < still.mean() < 1.0
            # against the independent pointer-octree restatement of src/piecewise_icp.py:17-235 (oracle/piecewise_octree.py):
            # the same rows in the same order (centroid sums differ in their last bits: 1e-9)
            from oracle import piecewise_octree as PO
            ref = PO.piecewise_icp(c["src"].astype(np.float64), c["tgt"].astype(np.float64), 1.4, 10, "brienz_tls")
            assert ref["dvfs"].shape == dvfs.shape
            assert np.abs(ref["dvfs"] - dvfs).max() < 1e-9 and np.abs(ref["dvfms"] - dvfms).max() < 1e-9
            assert np.abs(ref["visualize"] - vis).max() < 1e-9
        else:
            assert np.median(dvfms[:, 3]) < 0.2


def test_kabsch2_mirror_vs_golden_and_oracle(golden_dir):
    """src/functions.py:12-85 on the GPU against vectors produced by the reference's own torch function."""
    from fusion4landslide_amd import engine
    from fusion4landslide_amd.src.functions import kabsch_transformation_estimation
    g = np.load(os.path.join(golden_dir, "kabsch_golden.npz"))
    for j in (0, 1):
        x1, x2, w = (torch.from_numpy(g[f"k2_{j}_{k}"]).cuda() for k in ("x1", "x2", "w"))
        R, t, res, flag = kabsch_transformation_estimation(x1, x2, w)
        assert flag is False and R.shape == g[f"k2_{j}_R"].shape and t.shape == g[f"k2_{j}_t"].shape
        assert np.abs(R.cpu().numpy() - g[f"k2_{j}_R"]).max() < 1e-9
        assert np.abs(t.cpu().numpy() - g[f"k2_{j}_t"]).max() < 1e-9
        assert np.abs(res.cpu().numpy() - g[f"k2_{j}_res"]).max() < 1e-9
        R, t, _, _ = kabsch_transformation_estimation(x1, x2, None)
        assert np.abs(R.cpu().numpy() - g[f"k2_{j}_R_now"]).max() < 1e-9
        assert np.abs(t.cpu().numpy() - g[f"k2_{j}_t_now"]).max() < 1e-9
    # float32 batch, threshold on the NORMALISED weights, against the numpy restatement
    rng = np.random.default_rng(3)
    x1 = rng.uniform(-1, 1, (7, 300, 3)).astype(np.float32)
    x2 = (x1 @ np.array([[0.999, -0.04, 0.0], [0.04, 0.999, 0.0], [0.0, 0.0, 1.0]], np.float32) + 0.1).astype(np.float32)
    w = rng.uniform(0, 1, (7, 300)).astype(np.float32)
    for thr in (0, 1.0 / 300):
        R, t, res, _ = kabsch_transformation_estimation(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(),
                                                        torch.from_numpy(w).cuda(), w_threshold=thr)
        Rr, tr = O.kabsch_transformation_estimation(x1, x2, w, w_threshold=thr)
        assert np.abs(R.cpu().numpy() - Rr).max() < 2e-5 and np.abs(t.cpu().numpy() - tr).max() < 2e-5
    with pytest.raises(NotImplementedError):
        kabsch_transformation_estimation(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), best_k=10)


def test_median_resolution_vs_kdtree():
    """_compute_median_resolution (src/coarse_to_fine_matching_base.py:2716-2754): sklearn kd-tree 2-NN there, scipy here."""
    from scipy.spatial import cKDTree
    from fusion4landslide_amd import engine, synthetic
    c = synthetic.two_epoch_cloud(40_000, 9, 1.386, seed=5)
    ref = []
    for pts in (c["src"], c["tgt"][:30_001]):
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=2)
        ref.append(np.median(d[:, 1]))
    got = engine.median_resolution(torch.from_numpy(c["src"]).cuda(), torch.from_numpy(c["tgt"][:30_001]).cuda())
    assert abs(got - max(ref)) <= 1e-12 * max(ref) + 1e-15
    assert abs(engine.median_resolution(torch.from_numpy(c["src"]).cuda()) - ref[0]) <= 1e-12
    # the partition's own neighbour search hands out the same nearest-neighbour distances (f4l_knn_normals_nn1): bit-equal to
    # the 2-NN pass, the same rows and normals as without that output, and the same median when it stands in for the 2-NN pass
    s_dev, t_dev = torch.from_numpy(c["src"]).cuda(), torch.from_numpy(c["tgt"][:30_001]).cuda()
    idx, nrm, nn1 = engine.knn_normals(s_dev, 30, return_nn1=True)
    idx0, nrm0 = engine.knn_normals(s_dev, 30)
    assert torch.equal(idx, idx0) and torch.equal(nrm, nrm0)
    assert torch.equal(nn1, engine.knn(s_dev, 2, return_d2=True)[1][:, 1])
    assert engine.median_resolution(s_dev, t_dev, src_nn1_d2=nn1) == got
    # ... also where the lane-per-query kernel hands queries to the wave-per-query search (sparse outliers, duplicates)
    rng = np.random.default_rng(3)
    odd = np.concatenate([c["src"][:5000], c["src"][:40], rng.uniform(-50, 80, (60, 3)).astype(np.float32)])
    odd_dev = torch.from_numpy(odd).cuda()
    assert torch.equal(engine.knn_normals(odd_dev, 30, return_nn1=True)[2], engine.knn(odd_dev, 2, return_d2=True)[1][:, 1])


def test_median_f64_is_numpys_median():
    """f4l_median_f64 (the last step of _compute_median_resolution): numpy.median of strided doubles, odd and even counts,
    duplicates, negative values, a single element -- bit for bit."""
    import ctypes as C
    from fusion4landslide_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(3)
    for n, stride in ((1, 1), (2, 1), (7, 1), (10_000, 1), (9_999, 2), (250_001, 3)):
        v = rng.normal(size=(n, stride)) * np.where(rng.uniform(size=(n, 1)) < 0.2, 0.0, 1.0)  # (a fifth exact zeros: ties)
        d = torch.from_numpy(v).cuda()
        out = torch.empty((1,), dtype=torch.float64, device="cuda")
        nbytes = lib().f4l_median_f64_workspace_bytes(n)
        ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device="cuda")
        col = stride - 1
        check(lib().f4l_median_f64(d.data_ptr() + 8 * col, n, stride, ptr(out), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_median_f64")
        assert float(out.item()) == float(np.median(v[:, col])), (n, stride)
    # what the path feeds it: millions of positive values within a few octaves (all but the last digits of the select shared)
    for n in (3_000_001, 2_000_000):
        v = 0.03 + 0.02 * np.abs(rng.normal(size=n))
        v[: n // 3] = v[0]  # (a third of them one value: the median may sit in the run)
        d = torch.from_numpy(v).cuda()
        out = torch.empty((1,), dtype=torch.float64, device="cuda")
        nbytes = lib().f4l_median_f64_workspace_bytes(n)
        ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device="cuda")
        check(lib().f4l_median_f64(ptr(d), n, 1, ptr(out), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_median_f64")
        assert float(out.item()) == float(np.median(v)), n


def test_nn_query_vs_kdtree():
    """f4l_nn_query (the cKDTree(...).query of src/coarse_to_fine_matching_base.py:1042-1046): exact k nearest cloud
    points of queries that lie inside, at the border of and far outside the cloud's bounding box."""
    from scipy.spatial import cKDTree
    from fusion4landslide_amd import engine, synthetic
    rng = np.random.default_rng(41)
    c = synthetic.two_epoch_cloud(60_000, 9, 1.386, seed=6)
    cloud = c["src"]
    q = np.concatenate([c["tgt"][:20_000],                                         # another epoch of the same surface
                        cloud[:500],                                               # exact members of the cloud (d = 0)
                        rng.uniform(-30, 40, (300, 3)).astype(np.float32),         # anywhere, mostly outside the box
                        cloud[:200] + np.float32([0, 0, 25.0])])                   # far above
    tree = cKDTree(cloud.astype(np.float64))
    for k in (1, 4):
        idx, d2 = engine.nn_query(torch.from_numpy(cloud).cuda(), torch.from_numpy(q).cuda(), k, return_d2=True)
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
        dref, iref = tree.query(q.astype(np.float64), k=k)
        dref, iref = dref.reshape(len(q), k), iref.reshape(len(q), k)
        assert np.abs(np.sqrt(d2) - dref).max() <= 1e-12 * max(1.0, dref.max())
        same = idx == iref
        # index mismatches only inside exact-distance tie groups
        assert (np.abs(np.sqrt(d2) - dref)[~same] <= 1e-12).all()
        assert same.mean() > 0.999
        # distances really are those of the returned indices
        got = ((cloud[idx.reshape(-1)].astype(np.float64) - np.repeat(q.astype(np.float64), k, axis=0)) ** 2).sum(1)
        assert np.abs(got - d2.reshape(-1)).max() <= 1e-12 * max(1.0, d2.max())
    # the per-cell table of the whole grid (two loads per row of cells) against the binary searches over the occupied cells
    import os
    os.environ["F4L_KNN_NO_DENSE"] = "1"
    try:
        idx2, d22 = engine.nn_query(torch.from_numpy(cloud).cuda(), torch.from_numpy(q).cuda(), 4, return_d2=True)
    finally:
        del os.environ["F4L_KNN_NO_DENSE"]
    assert (idx2.cpu().numpy() == idx).all() and (d22.cpu().numpy() == d2).all()
    # k <= 4 runs one lane per query on its own block (nn_small_kernel): the same answers as the wave-shared candidate sets
    for k in (1, 4):
        i1, d1 = engine.nn_query(torch.from_numpy(cloud).cuda(), torch.from_numpy(q).cuda(), k, return_d2=True)
        os.environ["F4L_KNN_NO_SMALL"] = "1"
        try:
            i1w, d1w = engine.nn_query(torch.from_numpy(cloud).cuda(), torch.from_numpy(q).cuda(), k, return_d2=True)
        finally:
            del os.environ["F4L_KNN_NO_SMALL"]
        assert torch.equal(i1, i1w) and torch.equal(d1, d1w)
    # a single query, a single-point cloud, no queries
    one = engine.nn_query(torch.from_numpy(cloud[:1]).cuda(), torch.from_numpy(q[:7]).cuda(), 1)
    assert (one.cpu().numpy() == 0).all()
    assert engine.nn_query(torch.from_numpy(cloud).cuda(), torch.zeros((0, 3)).cuda(), 1).shape == (0, 1)


def test_nn_query_small_k_is_the_ordered_brute_force_on_ties_and_displaced_queries():
    """The small-k walk (k <= 4: float32 scan, exact re-measurement of the survivors, exact walk on near-ties, later rounds clipped
    to the ball of the known k-th distance) against the definition: the k smallest by (double d2 from the float coordinates, index)
    -- on a lattice (every distance tied many times), with duplicated cloud points, queries on lattice points, between them,
    displaced by several cells and far outside the cloud."""
    from fusion4landslide_amd import engine
    rng = np.random.default_rng(101)
    gx, gy = np.meshgrid(np.arange(60, dtype=np.float32) * 0.25, np.arange(50, dtype=np.float32) * 0.25)
    cloud = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32)], axis=1)
    cloud = np.concatenate([cloud, cloud[::7], rng.uniform(0, 12, (500, 3)).astype(np.float32) * np.float32([1, 1, 0.02])])  # + duplicates + scatter
    q = np.concatenate([cloud[rng.choice(len(cloud), 400)],                                   # on cloud points (d2 = 0, ties with duplicates)
                        cloud[rng.choice(len(cloud), 400)] + np.float32([0.125, 0.125, 0]),   # cell centres: four-way ties
                        cloud[rng.choice(len(cloud), 400)] + np.float32([0.0, 0.0, 0.9]),     # displaced across cells (flat grid: z is free)
                        rng.uniform(-5, 20, (300, 3)).astype(np.float32),                     # anywhere, also far outside
                        rng.uniform(0, 12, (500, 3)).astype(np.float32) * np.float32([1, 1, 0.02])])
    c64, q64 = cloud.astype(np.float64), q.astype(np.float64)
    d2_all = ((q64[:, None, 0] - c64[None, :, 0]) ** 2 + (q64[:, None, 1] - c64[None, :, 1]) ** 2) + (q64[:, None, 2] - c64[None, :, 2]) ** 2
    order = np.lexsort((np.broadcast_to(np.arange(len(cloud)), d2_all.shape), d2_all), axis=1)  # by d2, then index
    for k in (1, 2, 3, 4):
        idx, d2 = engine.nn_query(torch.from_numpy(cloud).cuda(), torch.from_numpy(q).cuda(), k, return_d2=True)
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
        ref = order[:, :k]
        assert np.array_equal(idx, ref), (k, int((idx != ref).any(axis=1).sum()))
        assert np.array_equal(d2, np.take_along_axis(d2_all, ref, axis=1))


def test_epoch_join_is_the_two_searches_it_replaces():
    """f4l_epoch_join: the second epoch binned once for its two searches.  Bit-equal to f4l_knn(tgt, 2)[:, 1] (what
    `_compute_median_resolution`, src/coarse_to_fine_matching_base.py:2716-2754, takes the median of) and to
    f4l_nn_query(src, tgt, 1) (the label transfer), both checked against a KD-tree above; targets inside, at the border of
    and far outside the source's box; f4l_labels_to_csr_via = f4l_labels_to_csr of the gathered labels."""
    from scipy.spatial import cKDTree
    from fusion4landslide_amd import engine, synthetic
    rng = np.random.default_rng(43)
    c = synthetic.two_epoch_cloud(80_000, 9, 1.386, seed=8)
    src = c["src"]
    tgt = np.concatenate([c["tgt"][:50_000], src[:300], rng.uniform(-30, 40, (300, 3)).astype(np.float32),
                          src[:200] + np.float32([0, 0, 25.0]), c["tgt"][:7]])      # (exact duplicates inside tgt: nn1 = 0)
    s_d, t_d = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    idx, nn1 = engine.epoch_join(s_d, t_d)
    _, d2 = engine.knn(t_d, 2, return_d2=True)
    assert torch.equal(nn1, d2[:, 1].contiguous())
    assert torch.equal(idx, engine.nn_query(s_d, t_d, 1)[:, 0].contiguous())
    dref, _ = cKDTree(src.astype(np.float64)).query(tgt.astype(np.float64), k=1)
    got = np.sqrt(((src[idx.cpu().numpy()].astype(np.float64) - tgt.astype(np.float64)) ** 2).sum(1))
    assert np.abs(got - dref).max() <= 1e-12 * max(1.0, dref.max())
    assert engine.median_resolution(s_d, t_d) == engine.median_resolution(s_d, t_d, tgt_nn1_d2=nn1)
    idx_only, none = engine.epoch_join(s_d, t_d, return_nn1=False)
    assert none is None and torch.equal(idx_only, idx)
    # labels through the join
    K = 53
    labels = rng.integers(0, K, len(src)).astype(np.int32)
    labels[rng.choice(len(src), 500, replace=False)] = -1
    l_d = torch.from_numpy(labels).cuda()
    o1, f1 = engine.labels_to_csr(l_d[idx.to(torch.int64)], K)
    o2, f2 = engine.labels_to_csr_via(l_d, idx, K)
    assert torch.equal(o1, o2) and torch.equal(f1, f2)
    via = idx.clone()
    via[::97] = -1
    via[5::101] = len(src) + 3                                                       # outside the labelled cloud: no patch
    lab_of = np.where((via.cpu().numpy() >= 0) & (via.cpu().numpy() < len(src)), labels[np.clip(via.cpu().numpy(), 0, len(src) - 1)], -1)
    o3, f3 = engine.labels_to_csr_via(l_d, via, K)
    o4, f4 = engine.labels_to_csr(torch.from_numpy(lab_of.astype(np.int32)).cuda(), K)
    assert torch.equal(o3, o4) and torch.equal(f3, f4)
    # two points per cloud, one target
    i2, n2 = engine.epoch_join(s_d[:2], t_d[:2])
    assert i2.shape == (2,) and n2.shape == (2,) and abs(float(n2[0]) - float(((tgt[0].astype(np.float64) - tgt[1]) ** 2).sum())) <= 1e-12
    with pytest.raises(Exception):
        engine.epoch_join(s_d, t_d[:1])                                              # (no "other" target point: EINVAL)
    i1, _ = engine.epoch_join(s_d, t_d[:1], return_nn1=False)
    assert int(i1[0]) == int(idx[0])


def test_voxel_downsample_and_subsampling_vs_oracle():
    """f4l_voxel_downsample against the numpy restatement of Open3D's voxel grid filter, and the whole
    `_voxel_subsampling` bookkeeping (src/coarse_to_fine_matching_base.py:1012-1057) against numpy + a KD-tree."""
    from scipy.spatial import cKDTree
    from fusion4landslide_amd import engine, synthetic
    from oracle import oracle as O
    c = synthetic.two_epoch_cloud(50_000, 9, 1.386, seed=7, origin=(2647.0, 1177.0, 1500.0))
    for voxel in (0.35, 0.05, 40.0):
        xyz = c["src"]
        pts, cnt, vop = engine.voxel_downsample(torch.from_numpy(xyz).cuda(), voxel, return_map=True)
        rp, rc, rv = O.voxel_downsample(xyz, voxel)
        assert pts.shape[0] == len(rp) and np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(vop.cpu().numpy(), rv)
        assert np.abs(pts.cpu().numpy() - rp).max() <= 1e-9
    out = engine.voxel_subsampling(torch.from_numpy(c["src"]).cuda(), torch.from_numpy(c["tgt"]).cuda())
    ref_res = max(np.median(cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=2)[0][:, 1]) for p in (c["src"], c["tgt"]))
    assert abs(out["voxel_size"] - ref_res) <= 1e-12
    for name in ("src", "tgt"):
        xyz, o = c[name], out[name]
        rp, _, _ = O.voxel_downsample(xyz, out["voxel_size"])
        sub = o["pts_sub"].cpu().numpy()
        assert sub.dtype == np.float32 and np.array_equal(sub, rp.astype(np.float32))
        d, i = cKDTree(xyz.astype(np.float64)).query(sub.astype(np.float64), k=1)
        v2p = o["idx_voxel2pts"].cpu().numpy()
        dd = np.sqrt(((xyz[v2p].astype(np.float64) - sub.astype(np.float64)) ** 2).sum(1))
        # (the centre of a two-point voxel is equidistant from both: index ties are common, distances must agree)
        assert np.abs(dd - d).max() <= 1e-12 and (v2p == i).mean() > 0.95
        p2v = o["idx_pts2voxel"].cpu().numpy()
        assert p2v.shape == (len(xyz),) and (p2v >= -1).all()
        hit = p2v >= 0
        assert np.array_equal(np.sort(np.unique(v2p)), np.nonzero(hit)[0])
        assert (v2p[p2v[hit]] == np.nonzero(hit)[0]).all()


def test_robust_rigid_fit_vs_per_set_restatement():
    """`filter_input` after the network (src/models/outlier_classifier.py:70-103) for many ragged sets at once, against
    the oracle's Kabsch #2 applied set by set with the same decisions."""
    from fusion4landslide_amd.src.functions import robust_rigid_fit
    from oracle import oracle as O
    rng = np.random.default_rng(51)
    sizes = [40, 7, 3, 120, 5, 64, 0, 33]
    corr, w = [], []
    for k, m in enumerate(sizes):
        x1 = rng.normal(0, 2.0, (m, 3))
        R0 = rot_from_axis_angle(rng.normal(size=3), 0.05 * (k + 1))
        x2 = x1 @ R0.T + rng.normal(0, 0.3, 3) + rng.normal(0, 0.01, (m, 3))
        bad = rng.random(m) < 0.2
        x2[bad] += rng.normal(0, 1.5 if k != 3 else 30.0, (int(bad.sum()), 3))     # set 3: gross outliers
        corr.append(np.c_[x1, x2])
        w.append(np.where(bad, rng.uniform(0, 0.3, m), rng.uniform(0.7, 1.0, m)))
    off = np.zeros(len(sizes) + 1, np.int64); np.cumsum(sizes, out=off[1:])
    corr, w = np.concatenate(corr), np.concatenate(w)
    for coeff in (1.0, 2.5):
        out = robust_rigid_fit(torch.from_numpy(corr).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(w).cuda(), coeff)
        for p, m in enumerate(sizes):
            if m == 0:
                assert not bool(out["robust_estimate"][p])
                continue
            c, ww = corr[off[p]:off[p + 1]], w[off[p]:off[p + 1]]
            R, t = O.kabsch_transformation_estimation(c[None, :, :3], c[None, :, 3:], ww[None])
            res = np.linalg.norm(c[:, :3] @ R[0].T + t[0].T - c[:, 3:], axis=1)
            med = np.sort(res)[(m - 1) // 2]                                        # torch.median: lower median
            inl = res < coeff * med
            robust = inl.sum() >= 5 and med < 0.5
            assert bool(out["robust_estimate"][p]) == robust, p
            assert np.array_equal(out["inliers"][off[p]:off[p + 1]].cpu().numpy(), inl), p
            if robust:
                R, t = O.kabsch_transformation_estimation(c[None, :, :3], c[None, :, 3:], inl[None].astype(np.float64))
            assert np.abs(out["rot_est"][p].cpu().numpy() - R[0]).max() <= 1e-9, p
            assert np.abs(out["trans_est"][p].cpu().numpy() - t[0]).max() <= 1e-9, p


def test_supervoxel_device_assisted_sweeps_give_the_same_labels(monkeypatch):
    """f4l_supervoxel computes two sweeps of the segmentation on the GPU (starting lambda's per-point minimum metric,
    boundary flags + distance to the representative after the fusion); F4L_SV_HOST_ONLY=1 keeps them on the host.  The
    metric must be bit-identical on both sides: same labels, same count, on a cloud large enough to hit every branch."""
    from fusion4landslide_amd import engine, synthetic
    c = synthetic.two_epoch_cloud(250_000, 22, 1.386, seed=11, origin=(2647.0, 1177.0, 1500.0))
    xyz = torch.from_numpy(c["src"]).cuda()
    for res in (1.386, 0.4):
        lab, K = engine.supervoxel(xyz, 30, res)
        monkeypatch.setenv("F4L_SV_HOST_ONLY", "1")
        lab_h, K_h = engine.supervoxel(xyz, 30, res)
        monkeypatch.delenv("F4L_SV_HOST_ONLY")
        assert K == K_h and torch.equal(lab, lab_h)


def test_full_path_of_a_tile_end_to_end():
    """fusion4landslide_amd.pipeline.full_path (BASELINE configs[4] on one GPU): partition -> patches -> point matches ->
    Kabsch + ICP + rows -> refinement.  Structural checks, and the per-patch stage against the oracle on sampled patches (the
    patches come out of the pipeline, the oracle sees the same points)."""
    from fusion4landslide_amd import pipeline, synthetic
    c = synthetic.two_epoch_cloud(200_000, 20, 1.386, seed=3, roughness=0.05)
    src, tgt = torch.from_numpy(c["src"]).cuda(), torch.from_numpy(c["tgt"]).cuda()
    for partition in ("parallel", "identical"):
        r = pipeline.full_path(src, tgt, partition=partition)
        K = r["K"]
        if partition == "parallel":  # (the path shares one neighbour search between the resolution estimate and the partition)
            from fusion4landslide_amd import engine
            assert abs(r["resolution"] - np.sqrt(3.0) * 10.0 * engine.median_resolution(src, tgt)) <= 1e-12
            lab, K2 = engine.supervoxel_parallel(src, 30, r["resolution"])
            assert K2 == K and torch.equal(lab, r["labels"])
        assert r["labels"].shape == (200_000,) and int(r["labels"].max()) == K - 1 and 500 < K < 50_000
        assert r["rows"].shape == (200_000, 6) and r["T"].shape == (K, 4, 4) and set(r["stage_ms"]) >= {"supervoxel_partition", "patch_loop", "total"}
        order, so, to = r["order"].cpu().numpy(), r["src_off"].cpu().numpy(), r["tgt_off"].cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(200_000)) and so[-1] == 200_000 and to[-1] == 200_000
        rows = r["rows"].cpu().numpy()
        assert np.array_equal(rows[:, :3], c["src"][order])
        # displacement field: mostly within the planted motion's range, stable blocks well below the threshold
        mag = np.linalg.norm(rows[:, 3:] - rows[:, :3], axis=1)
        assert np.median(mag) < 0.12 and np.isfinite(mag).all()
        assert float(r["fitness"].mean()) > 0.5 and r["sparse"].shape[1] == 6 and r["sparse"].shape[0] > 50_000
