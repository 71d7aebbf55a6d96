"""GPU tests of the reference's OWN labels computed on the device (csrc/supervoxel_exact.hip, f4l_supervoxel_segment_exact): the
sequential fusion and the FIFO exchange of codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-248 as fixed points of
parallel passes.  The reference-held fixtures are in tests/test_gpu_parity.py::test_knn_normals_supervoxel_vs_golden and
tests/test_gpu_supervoxel_parallel.py::test_partition_against_the_large_reference_fixture (both call f4l_supervoxel, which takes
this path); here: random clouds against the one-core host replay of the same sequence (csrc/supervoxel_host.cpp, itself pinned by
those fixtures) -- every label, in orders that make the index-ordered dependency chains short (random) and long (sorted rows)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from fusion4landslide_amd import engine
    return engine


def _cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "surface":
        xy = rng.uniform(0, 30, (n, 2))
        p = np.c_[xy, 0.8 * np.sin(0.5 * xy[:, 0]) * np.cos(0.4 * xy[:, 1]) + rng.normal(0, 0.01, n)]
    elif kind == "rows":  # a voxel-filtered tile as PCL leaves it: rows along x, one after the other -- long chains of dependencies
        xy = rng.uniform(0, 30, (n, 2))
        p = np.c_[xy, 0.5 * np.sin(0.7 * xy[:, 0]) + rng.normal(0, 0.005, n)]
        p = p[np.lexsort((p[:, 0], np.floor(p[:, 1] / 0.2)))]
    elif kind == "volume":
        p = rng.uniform(0, 6, (n, 3))
    elif kind == "lattice":  # exactly equal distances everywhere, a few duplicated points
        m = int(round(n ** 0.5))
        gx, gy = np.meshgrid(np.arange(m) * 0.1, np.arange(m) * 0.1)
        p = np.c_[gx.ravel(), gy.ravel(), np.zeros(m * m)]
        p = np.r_[p, p[rng.integers(0, len(p), 7)]]
        p = p[rng.permutation(len(p))]
    else:  # georeferenced, two sheets
        xy = rng.uniform(0, 20, (n, 2))
        z = np.where(rng.random(n) < 0.5, 0.0, 0.6) + 0.05 * xy[:, 0]
        p = np.c_[xy, z] + np.array([2647000.0, 1177000.0, 1500.0]) % 4096
    return np.ascontiguousarray(p, dtype=np.float32)


@pytest.mark.parametrize("kind,n,k,res", [("surface", 30_000, 30, 1.0), ("surface", 120_000, 15, 0.6), ("rows", 60_000, 30, 0.8),
                                          ("volume", 25_000, 12, 0.9), ("lattice", 10_000, 8, 0.55), ("georef", 40_000, 30, 1.2),
                                          ("rows", 200_000, 30, 0.5), ("surface", 3_000, 30, 4.0)])
def test_device_labels_equal_the_host_replay(eng, kind, n, k, res, monkeypatch):
    import torch
    xyz = torch.from_numpy(_cloud(kind, n, seed=n + k)).cuda()
    monkeypatch.delenv("F4L_SV_EXACT_HOST", raising=False)
    lab_d, K_d = eng.supervoxel(xyz, k, res)
    monkeypatch.setenv("F4L_SV_EXACT_HOST", "1")
    lab_h, K_h = eng.supervoxel(xyz, k, res)
    assert K_d == K_h
    assert torch.equal(lab_d, lab_h), f"{int((lab_d != lab_h).sum())} of {len(lab_d)} labels differ"
    cnt = torch.bincount(lab_d.long(), minlength=K_d)
    assert int(cnt.min()) > 0 and int(lab_d.max()) == K_d - 1


def test_wide_queue_and_restart_give_the_same_labels(eng, monkeypatch):
    """The evaluation kernel has two shapes: 16 lanes and a queue of 256 nodes per representative (the normal one), 32 lanes and 1024
    (F4L_SV_EXACT_WIDE, and the automatic restart when a closure outgrows the narrow queue: a dense VOLUME at a coarse resolution,
    where a representative's closure visits several hundred others).  Both equal the host replay."""
    import ctypes as C
    import torch
    from fusion4landslide_amd._lib import lib, ptr, stream_ptr
    xyz = torch.from_numpy(_cloud("surface", 50_000, seed=9)).cuda()
    monkeypatch.setenv("F4L_SV_EXACT_HOST", "1")
    lab_h, K_h = eng.supervoxel(xyz, 30, 1.0)
    monkeypatch.delenv("F4L_SV_EXACT_HOST")
    monkeypatch.setenv("F4L_SV_EXACT_WIDE", "1")
    lab_w, K_w = eng.supervoxel(xyz, 30, 1.0)
    monkeypatch.delenv("F4L_SV_EXACT_WIDE")
    assert K_w == K_h and torch.equal(lab_w, lab_h)
    # a closure beyond the narrow queue: the entry restarts wide by itself (stats[4] = the largest closure of the run that finished)
    vol = torch.from_numpy(_cloud("volume", 150_000, seed=2)).cuda()
    knn, nrm = eng.knn_normals(vol, 40)
    n = vol.shape[0]
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    nb = lib().f4l_supervoxel_segment_exact_workspace_bytes(n, 40)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    K, stats = C.c_int32(0), (C.c_int32 * 5)()
    rc = lib().f4l_supervoxel_segment_exact(ptr(vol), ptr(nrm), ptr(knn), n, 40, 2.5, ptr(labels), C.byref(K), stats, ptr(ws), C.c_size_t(nb), stream_ptr())
    assert rc == 0 and stats[4] > 252, stats[4]   # (beyond the narrow queue: this run was the wide one)
    monkeypatch.setenv("F4L_SV_EXACT_HOST", "1")
    lab_h2, K_h2 = eng.supervoxel(vol, 40, 2.5)
    assert K_h2 == K.value and torch.equal(labels, lab_h2)


def test_segment_exact_entry_reports_its_passes_and_refuses_what_it_cannot_hold(eng):
    """The C entry by itself: stats (lambda rounds, fusion passes, exchange generations and passes), and k = 1 -- whose pools cannot
    hold the cell-count hash set -- refused with F4L_EUNSUPPORTED (f4l_supervoxel then replays on the host)."""
    import ctypes as C
    import torch
    from fusion4landslide_amd._lib import lib, ptr, stream_ptr
    xyz = torch.from_numpy(_cloud("surface", 20_000, seed=3)).cuda()
    knn, nrm = eng.knn_normals(xyz, 20)
    n = xyz.shape[0]
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    nb = lib().f4l_supervoxel_segment_exact_workspace_bytes(n, 20)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    K, stats = C.c_int32(0), (C.c_int32 * 5)()
    rc = lib().f4l_supervoxel_segment_exact(ptr(xyz), ptr(nrm), ptr(knn), n, 20, 1.0, ptr(labels), C.byref(K), stats, ptr(ws), C.c_size_t(nb), stream_ptr())
    assert rc == 0 and K.value > 10
    assert 3 <= stats[0] <= 40 and stats[1] >= stats[0] and 1 <= stats[2] <= 50 and stats[3] >= stats[2] and 20 <= stats[4] <= 252
    lab_h, K_h = eng.supervoxel(xyz, 20, 1.0)
    assert K_h == K.value and torch.equal(labels, lab_h)
    knn1 = knn[:, :1].contiguous()
    nb1 = max(int(lib().f4l_supervoxel_segment_exact_workspace_bytes(n, 1)), 1)
    ws1 = torch.empty(nb1, dtype=torch.uint8, device="cuda")
    rc = lib().f4l_supervoxel_segment_exact(ptr(xyz), ptr(nrm), ptr(knn1), n, 1, 1.0, ptr(labels), C.byref(K), stats, ptr(ws1), C.c_size_t(nb1), stream_ptr())
    assert rc == -4 or rc != 0  # F4L_EUNSUPPORTED


def test_every_point_in_a_cell_of_its_own_follows_the_reference(eng, monkeypatch):
    """K == n (a resolution below the point spacing).  The reference's `--number == n_supervoxels` (:160) can no longer
    fire once a first absorption took the count below K, and its check after a centre's turn (:169) fires at once when the FIRST
    centre absorbs nothing: n singletons in one order of the points, ONE supervoxel in another.  The device path leaves this case to
    the host replay (f4l_supervoxel_segment_exact returns F4L_EUNSUPPORTED); f4l_supervoxel equals the live reference in both."""
    import ctypes as C
    import torch
    from fusion4landslide_amd._lib import lib, ptr, stream_ptr
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    gx, gy = np.meshgrid(np.arange(25) * 0.1, np.arange(25) * 0.1)
    p = (np.c_[gx.ravel(), gy.ravel(), np.zeros(625)] + rng.normal(0, 0.01, (625, 3))).astype(np.float32)
    monkeypatch.delenv("F4L_SV_EXACT_HOST", raising=False)
    for cloud, want in ((p, 625), (np.roll(p, -1, axis=0), 1)):
        xyz = torch.from_numpy(np.ascontiguousarray(cloud)).cuda()
        lab, K = eng.supervoxel(xyz, 8, 0.02)
        assert K == want and len(torch.unique(lab)) == want
        if O.have_ref():
            r = O.ref_supervoxel(cloud, 8, 0.02)
            assert r["n_grid_cells"] == 625 and r["n_supervoxels"] == want and np.array_equal(r["labels"], lab.cpu().numpy())
        else:
            r = O.supervoxel(cloud, 8, 0.02)
            assert r["n_supervoxels"] == want and np.array_equal(r["labels"], lab.cpu().numpy())
    knn, nrm = eng.knn_normals(xyz, 8)
    labels = torch.empty(625, dtype=torch.int32, device="cuda")
    nb = lib().f4l_supervoxel_segment_exact_workspace_bytes(625, 8)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    K = C.c_int32(0)
    rc = lib().f4l_supervoxel_segment_exact(ptr(xyz), ptr(nrm), ptr(knn), 625, 8, 0.02, ptr(labels), C.byref(K), None, ptr(ws), C.c_size_t(nb), stream_ptr())
    assert rc == -4  # F4L_EUNSUPPORTED


def test_a_neighbour_graph_that_cannot_reach_the_target_count_is_reported(eng):
    """The reference's fusion loop (:117-176) has no exit when the neighbour graph has more connected components than the resolution
    grid has occupied cells -- lambda doubles for ever (found by tools/gpu/fuzz_supervoxel_exact.py: a thin strip, k = 4, a resolution
    of a few cells: the C oracle, which restates the loop, does not return on it either).  f4l_supervoxel returns
    F4L_EUNSUPPORTED instead (the host replay proves that no round can absorb anything any more), and a cloud whose graph is connected
    at the same k still gets its labels."""
    import torch
    from fusion4landslide_amd._lib import F4LError
    rng = np.random.default_rng(1003)
    # two clumps 100 m apart, 8 neighbours each: two components; one cell of the resolution grid holds both: K = 1 is out of reach
    a = rng.normal(0, 0.5, (60, 3)) * [1, 1, 0.05]
    p = np.r_[a, a[::-1] + [100.0, 0, 0]].astype(np.float32)
    with pytest.raises(F4LError):
        eng.supervoxel(torch.from_numpy(p).cuda(), 8, 1000.0)
    # (the same two clumps with a cell each: K = 2 is reached, every clump one supervoxel)
    lab, K = eng.supervoxel(torch.from_numpy(p).cuda(), 8, 50.0)
    lab = lab.cpu().numpy()
    assert K == 2 and len(set(lab[:60])) == 1 and len(set(lab[60:])) == 1 and lab[0] != lab[60]
