"""GPU tests of the supervoxel segmentation on the device (f4l_supervoxel_segment_device / f4l_supervoxel_parallel, rows
a5-a7 without the host).  The parallel variant is deterministic: the device's labels must equal those of its independent
numpy restatement (oracle/sv_parallel.py) EXACTLY; what ties it to the REFERENCE are numbers the reference's own code produced
(tests/golden/supervoxel_*.npz): K = its GridSample count and the starting lambda of its fusion, both bit-exact; the quality of
the device's partition against its labels; the invariants (labels 0..K-1 non-empty, fixed point of the boundary exchange); and
the per-patch displacements a piecewise motion yields on its partition and on the device's."""
import glob
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402
from oracle import sv_parallel as M  # noqa: E402
from tests._util import partition_quality, piecewise_motion_scene  # noqa: E402

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "supervoxel_*.npz")))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from fusion4landslide_amd import engine
    return engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_device_segmentation_equals_its_numpy_model(eng, path):
    g = np.load(path)
    xyz, knn, nrm, res = g["xyz"], g["knn_idx"], g["normals"], float(g["resolution"])
    labels, info, reps = eng.supervoxel_segment_device(dev(xyz), dev(nrm), dev(knn.astype(np.int32)), res, return_reps=True)
    info = info.cpu().numpy()
    ref = M.segment(xyz, nrm, knn, res)
    assert info[0] == ref["n_supervoxels"] == int(g["n_grid_cells"]) and info[1] == ref["K_target"] and info[2] == 0
    # K and the fusion's starting lambda are REFERENCE-held numbers: the count of the reference's own GridSample and the value its
    # own Median gives over its own metric (supervoxel_segmentation.h:105-113, 254-265; tools/make_golden_supervoxel.py): bit-exact
    assert info[1] == int(g["n_grid_cells"]) == int(g["n_supervoxels"])
    assert eng.supervoxel_lambda0(info) == float(g["lambda0"]) == ref["lambda0"]
    assert info[3] == ref["sweeps"] and info[6] == ref["rounds"]
    assert np.array_equal(labels.cpu().numpy(), ref["labels"])
    assert np.array_equal(reps.cpu().numpy()[:info[0]], ref["reps"])
    # the quality of the DEVICE's partition against the labels the reference's own code produced on this cloud
    rms_ref, dev_ref, cv_ref = partition_quality(xyz, nrm, g["labels"])
    rms, dvn, cv = partition_quality(xyz, nrm, labels.cpu().numpy())
    assert rms <= 1.10 * rms_ref and dvn <= 1.15 * dev_ref + 1e-4 and cv <= 1.3 * cv_ref, ((rms, dvn, cv), (rms_ref, dev_ref, cv_ref))
    inv = M.check_invariants(xyz, nrm, knn.astype(np.int64), res, labels.cpu().numpy(), reps.cpu().numpy()[:info[0]])
    assert inv["K_equals_cells"] and inv["labels_contiguous"] and inv["all_non_empty"] and inv["fixed_point_violations"] == 0
    # run-to-run identical
    again, _ = eng.supervoxel_segment_device(dev(xyz), dev(nrm), dev(knn.astype(np.int32)), res)
    assert torch.equal(again, labels)
    # ... and the same however the passes are split between the schedule of launches and the kernels that run what the
    # schedule did not cover: everything as launches, none of it, and two rounds and one
    # sweep as launches with all the rest inside the one-workgroup kernels
    for name, value in (("F4L_SV_LAUNCHES", "1"), ("F4L_SV_SCHEDULED", "2,1"), ("F4L_SV_SCHEDULED", "0,0")):
        os.environ[name] = value
        try:
            other, info2 = eng.supervoxel_segment_device(dev(xyz), dev(nrm), dev(knn.astype(np.int32)), res)
        finally:
            del os.environ[name]
        assert torch.equal(other, labels) and np.array_equal(info2.cpu().numpy(), info), (name, value)


def test_partition_against_the_large_reference_fixture(eng, golden_dir):
    """tests/golden/sv_large_ref.npz: what the REFERENCE's own templates compute on a 300 k-point cloud (15 x the largest small
    fixture; labels, K, grid cells, lambda0, checksums of its neighbour lists -- tools/make_golden_supervoxel.py --large; the
    cloud comes back from its seed).  f4l_supervoxel (kNN + normals on the device, the reference's visiting order replayed)
    reproduces EVERY label; f4l_supervoxel_parallel has K and lambda0 bit-exact and its partition's quality within the bounds
    of the small fixtures.  Where the reference-compiled checker travelled with the snapshot (oracle/_ref/libf4l_ref.so, built
    in the build container by oracle/Makefile), a FRESH cloud of the same size is also put through the live reference here."""
    from tests._util import LARGE_CASE, bits_checksum, large_surface_cloud
    g = np.load(os.path.join(golden_dir, "sv_large_ref.npz"))
    c = LARGE_CASE
    xyz = large_surface_cloud(c["seed"], c["n"], c["extent"])
    assert bits_checksum(xyz) == int(g["xyz_checksum"]), "the generator no longer reproduces the fixture's cloud"
    idx, d2 = eng.knn(dev(xyz), c["k"], return_d2=True)
    assert bits_checksum(d2.cpu().numpy()) == int(g["knn_d2_checksum"])      # every squared distance, bit for bit
    assert bits_checksum(idx.cpu().numpy()) == int(g["knn_idx_checksum"])    # (random floats: no exact ties to reorder)
    labels, K = eng.supervoxel(dev(xyz), c["k"], c["resolution"])
    assert K == int(g["n_supervoxels"]) == int(g["n_grid_cells"])
    assert np.array_equal(labels.cpu().numpy(), g["labels"]), "labels differ from the reference's"
    par, Kp, knn, nrm, reps, info = eng.supervoxel_parallel(dev(xyz), c["k"], c["resolution"], return_intermediates=True)
    info = info.cpu().numpy() if hasattr(info, "cpu") else np.asarray(info)
    assert Kp == int(g["n_grid_cells"]) and int(info[1]) == int(g["n_grid_cells"]) and int(info[2]) == 0
    assert eng.supervoxel_lambda0(info) == float(g["lambda0"])
    nrm_h = nrm.cpu().numpy()
    assert np.allclose(np.abs(nrm_h).sum(axis=0), g["normals_abs_sum"], rtol=1e-9)
    rms_ref, dev_ref, cv_ref = partition_quality(xyz, nrm_h, g["labels"])
    rms, dvn, cv = partition_quality(xyz, nrm_h, par.cpu().numpy())
    assert rms <= 1.10 * rms_ref and dvn <= 1.15 * dev_ref + 1e-4 and cv <= 1.3 * cv_ref, ((rms, dvn, cv), (rms_ref, dev_ref, cv_ref))
    if O.have_ref():  # the live reference, on a cloud no fixture has seen
        seed = c["seed"] + int(time.time()) % 1000 + 1
        fresh = large_surface_cloud(seed, 200_000, 4.0)
        r = O.ref_supervoxel(fresh, c["k"], c["resolution"])
        lab2, K2 = eng.supervoxel(dev(fresh), c["k"], c["resolution"])
        assert K2 == r["n_supervoxels"] and np.array_equal(lab2.cpu().numpy(), r["labels"])
        par2, Kp2, _, nrm2, _, info2 = eng.supervoxel_parallel(dev(fresh), c["k"], c["resolution"], return_intermediates=True)
        info2 = info2.cpu().numpy() if hasattr(info2, "cpu") else np.asarray(info2)
        assert Kp2 == r["n_grid_cells"]
        # (lambda0 is the metric of ONE pair of points, the median: bit-equal when the device's normal of that pair equals the
        #  reference's to the last bit -- the fixtures' case --; the normals are only pinned to 1e-9, so on a cloud drawn from the
        #  clock the last bits of 1 - |n . n'| may differ: seen once in six rounds, 1.1e-16 absolute at seed-of-the-day)
        l0, l0_ref = eng.supervoxel_lambda0(info2), O.ref_lambda0(fresh, r["normals"], r["knn_idx"], c["resolution"])
        assert abs(l0 - l0_ref) <= 1e-12 * l0_ref, (seed, l0, l0_ref)
        q_ref, q_par = partition_quality(fresh, r["normals"], r["labels"]), partition_quality(fresh, r["normals"], par2.cpu().numpy())
        assert q_par[0] <= 1.10 * q_ref[0] and q_par[1] <= 1.15 * q_ref[1] + 1e-4 and q_par[2] <= 1.3 * q_ref[2], (q_par, q_ref)


def test_whole_partition_on_the_device_and_edge_cases(eng):
    rng = np.random.default_rng(5)
    xyz = np.c_[rng.uniform(0, 6, (30_000, 2)), np.zeros(30_000)]
    xyz[:, 2] = 0.3 * np.sin(xyz[:, 0]) * np.cos(1.3 * xyz[:, 1]) + rng.normal(0, 0.003, 30_000)
    xyz = xyz.astype(np.float32)
    labels, K, knn, nrm, reps, info = eng.supervoxel_parallel(dev(xyz), 30, 0.5, return_intermediates=True)
    ref = M.segment(xyz, nrm.cpu().numpy(), knn.cpu().numpy(), 0.5)
    assert K == ref["n_supervoxels"] == M.occupied_cells(xyz, 0.5) and int(info[2]) == 0
    assert np.array_equal(labels.cpu().numpy(), ref["labels"]) and np.array_equal(reps.cpu().numpy(), ref["reps"])
    # one cell: one supervoxel
    labels, K = eng.supervoxel_parallel(dev(xyz[:500]), 8, 100.0)
    assert K == 1 and bool((labels == 0).all())
    # disconnected neighbour graph with a target below the number of components: stops, and says so (status bit 0)
    a = rng.uniform(0, 1, (200, 3))
    two = np.concatenate([a, a + [50.0, 0, 0]]).astype(np.float32)
    labels, K, knn, nrm, reps, info = eng.supervoxel_parallel(dev(two), 8, 100.0, return_intermediates=True)
    assert K == 2 and int(info[1]) == 1 and int(info[2]) & 1
    assert not set(labels.cpu().numpy()[:200]) & set(labels.cpu().numpy()[200:])


def test_parallel_partition_recovers_a_piecewise_motion_like_the_reference_partition(eng):
    """Downstream tie of the device partition to the reference: the golden cloud's second epoch under
    a PIECEWISE rigid motion (tests/_util.piecewise_motion_scene: 16 blocks, each its own rotation <= 0.5 deg and translation
    <= 4 mm), cut into patches by the reference's own labels and by the DEVICE's parallel labels; the same per-patch ICP
    (f4l_piecewise_icp) on both, against the planted field.  Patches that straddle a block boundary cannot follow both blocks,
    so the error depends on where a partition puts its boundaries (p95 is millimetres where the median is 0.16 mm); the two
    error distributions must agree: median within 15 % + 0.02 mm, p95 within 15 %."""
    g = np.load([c for c in CASES if "surf_s4_n20000" in c][0])
    xyz, res = g["xyz"], float(g["resolution"])
    tgt, truth = piecewise_motion_scene(xyz)
    par, K = eng.supervoxel_parallel(dev(xyz), int(g["k"]), res)
    stats = []
    for lab, nsv in ((g["labels"], int(g["n_supervoxels"])), (par.cpu().numpy(), K)):
        order, off = eng.labels_to_csr(dev(lab.astype(np.int32)), nsv)
        s = eng.gather_points(dev(xyz), order)
        t = eng.gather_points(dev(tgt), order)  # the same points one epoch later: the same partition of the target
        out = eng.piecewise_icp(s, off, t, off, max_corr_dist=0.02, max_iter=30)
        rows = eng.apply_transform(s, off, out["T"]).cpu().numpy()
        est = rows[:, 3:].astype(np.float64) - rows[:, :3].astype(np.float64)
        err = np.linalg.norm(est - truth[order.cpu().numpy()], axis=1)
        stats.append((float(np.median(err)), float(np.quantile(err, 0.95))))
    (med_ref, p95_ref), (med_par, p95_par) = stats
    assert p95_ref > 5 * med_ref, stats  # the scene does depend on the partition
    assert med_par <= 1.15 * med_ref + 2e-5 and p95_par <= 1.15 * p95_ref, stats
    print(f"piecewise-motion scene: reference partition median {1e3 * med_ref:.3f} mm p95 {1e3 * p95_ref:.3f} mm; device partition "
          f"median {1e3 * med_par:.3f} mm p95 {1e3 * p95_par:.3f} mm")


def test_device_segmentation_1M_invariants_and_no_host_round_trip(eng):
    """A 1 M-point tile: invariants at full size (checked with torch on the device), and the stage really is asynchronous --
    the call returns before the device has finished (events recorded around it are not yet complete)."""
    from fusion4landslide_amd import synthetic
    d = synthetic.make_patches_device(1_000_000, 45, 1.386, torch.device("cuda"), seed=0)
    xyz = d["src"]
    k, res = 30, 1.386
    knn = eng.knn(xyz, k)
    nrm = eng.normals(xyz, knn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    labels, info, reps = eng.supervoxel_segment_device(xyz, nrm, knn, res, return_reps=True)
    e1.record()
    still_running = not e1.query()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    info = info.cpu().numpy()
    K = int(info[0])
    assert still_running, "the call returned only after the device had finished: something synchronised"
    assert info[2] == 0 and K == info[1] == M.occupied_cells(xyz.cpu().numpy(), res)
    cnt = torch.bincount(labels.to(torch.int64), minlength=K)
    assert cnt.shape[0] == K and int(cnt.min()) > 0 and int(labels.min()) == 0
    r = reps[:K].to(torch.int64)
    assert bool((labels[r].to(torch.int64) == torch.arange(K, device="cuda")).all()) and bool((r[1:] > r[:-1]).all())
    # fixed point of the exchange, on a sample of 50 k points against numpy
    pick = np.random.default_rng(0).choice(1_000_000, 50_000, replace=False)
    xh, nh, kh, lh, rh = xyz.cpu().numpy(), nrm.cpu().numpy(), knn.cpu().numpy().astype(np.int64), labels.cpu().numpy().astype(np.int64), r.cpu().numpy()
    dis = M.metric(xh, nh, pick, rh[lh[pick]], res)
    for j in range(k):
        b = lh[kh[pick, j]]
        diff = b != lh[pick]
        dd = M.metric(xh, nh, pick[diff], rh[b[diff]], res)
        assert not (dd < dis[diff]).any()
    # >= 20x over the 1.9 s of the sequential host stage: well under 95 ms
    assert ms < 95.0, ms
    print(f"device segmentation of 1 M points: {ms:.1f} ms, K = {K}, sweeps = {info[3]}")


def test_slab_pipeline_single_rank_equals_the_whole_partition(eng):
    """fusion4landslide_amd/slabs.py with its default (HIP) kernels at world_size 1: the multi-GPU partition pipeline reduces to
    f4l_supervoxel_parallel (the N > 1 orchestration is covered by tests/test_slabs_gloo.py)."""
    from fusion4landslide_amd import slabs
    rng = np.random.default_rng(8)
    xyz = np.c_[rng.uniform(0, 5, (20_000, 2)), np.zeros(20_000)].astype(np.float32)
    xyz[:, 2] = 0.2 * np.sin(2 * xyz[:, 0]) * np.cos(3 * xyz[:, 1])
    out = slabs.slab_supervoxel(dev(xyz), torch.arange(20_000, device="cuda"), 30, 0.5, None, 0, 1, halo=0.3)
    labels, K = eng.supervoxel_parallel(dev(xyz), 30, 0.5)
    assert out["K_total"] == out["K_local"] == K and out["n_uncertified"] == 0 and out["offset"] == 0
    assert torch.equal(out["labels"].to(torch.int32), labels) and torch.equal(out["gid"], torch.arange(20_000, device="cuda"))


def test_full_path_over_slabs_single_rank_equals_the_tile_path(eng):
    """pipeline.full_path_slabs (configs[4] spread over GPUs: slab partition, target epoch joined through the slabs, then the
    per-patch stages where the patches live) at world_size 1 against pipeline.full_path on the same tile: the same partition,
    patches, transforms and rows, bit for bit.  (N > 1: tests/test_slabs_gloo.py covers both exchanges.)"""
    from fusion4landslide_amd import pipeline, synthetic
    c = synthetic.two_epoch_cloud(60_000, 9, 1.386, seed=4)
    src, tgt = dev(c["src"]), dev(c["tgt"])
    res = 0.9
    whole = pipeline.full_path(src, tgt, resolution=res, max_iter=20, fixed_iters=True, partition="parallel")  # (the slab partition is the parallel variant)
    part = pipeline.full_path_slabs(src, torch.arange(src.shape[0], device="cuda"), tgt, None, 0, 1, halo=0.5, resolution=res, max_iter=20,
                                    fixed_iters=True)
    assert part["K_local"] == part["K_total"] == whole["K"] and part["offset"] == 0 and part["n_uncertified"] == 0
    assert torch.equal(part["gid"], whole["order"].to(torch.int64))
    for key in ("T", "fitness", "rmse", "iters", "rows", "src_off", "tgt_off"):
        assert torch.equal(part[key], whole[key]), key
    assert part["stage_ms"]["total"] > 0 and "target_exchange" in part["stage_ms"]


def test_whole_partition_is_capturable_into_a_hip_graph(eng):
    """f4l_supervoxel_parallel -- kNN, normals, segmentation -- only enqueues when its stream is under capture (f4l_knn then sizes
    its grid on the device instead of reading the bounding box and the cell count back): captured once, replayed, the labels
    of the eager call every time.  The same path on request (F4L_KNN_ASYNC) gives the eager path's neighbours exactly: the
    search is exact for any cell size."""
    rng = np.random.default_rng(9)
    xyz = np.c_[rng.uniform(0, 8, (60_000, 2)), np.zeros(60_000)]
    xyz[:, 2] = 0.3 * np.sin(xyz[:, 0]) * np.cos(1.3 * xyz[:, 1]) + rng.normal(0, 0.003, 60_000)
    x = dev(xyz.astype(np.float32))
    ref_labels, K = eng.supervoxel_parallel(x, 30, 0.5)
    idx_ref, d2_ref = eng.knn(x, 30, return_d2=True)
    os.environ["F4L_KNN_ASYNC"] = "1"
    try:
        idx, d2 = eng.knn(x, 30, return_d2=True)
        idx2 = eng.knn(x, 2)
    finally:
        del os.environ["F4L_KNN_ASYNC"]
    assert torch.equal(d2, d2_ref) and torch.equal(idx, idx_ref) and torch.equal(idx2, eng.knn(x, 2))
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.supervoxel_parallel(x, 30, 0.5, read_count=False)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            labels, info = eng.supervoxel_parallel(x, 30, 0.5, read_count=False)
    for _ in range(3):
        labels.zero_()
        info.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(labels, ref_labels) and int(info[0]) == K and int(info[2]) == 0
