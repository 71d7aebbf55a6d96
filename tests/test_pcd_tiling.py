"""The tiling front end (cpp_core/pcd_tiling/pcd_tiling.cpp -> fusion4landslide_amd/cpp_core/pcd_tiling/build/pcd_tiling.py).
PCL is not available here, so these are structural checks of the restated recursion (CPU) and of the GPU voxel filter
against its numpy restatement."""
import os

import numpy as np
import pytest

from fusion4landslide_amd.cpp_core.pcd_tiling.build import pcd_tiling as T
from fusion4landslide_amd.utils.ply import read_ply


def _write_cloud(path, xyz, rgb=None):
    T._write(path, T._Cloud(xyz, rgb))


def _surface(rng, n, x0, x1, y0, y1):
    xy = np.c_[rng.uniform(x0, x1, n), rng.uniform(y0, y1, n)]
    return np.c_[xy, 0.5 * np.sin(0.05 * xy[:, 0]) * np.cos(0.04 * xy[:, 1])].astype(np.float32)


def test_split_boxes_halve_the_longer_in_plane_side():
    lo, hi = np.array([0, 0, 0], np.float32), np.array([10, 4, 1], np.float32)
    (t1lo, t1hi, o1lo, o1hi), (t2lo, t2hi, o2lo, o2hi) = T._split_boxes(lo, hi, 2)
    assert t1lo.tolist() == [5, 0, 0] and t1hi.tolist() == [10, 4, 1]      # upper half first
    assert t2lo.tolist() == [0, 0, 0] and t2hi.tolist() == [5, 4, 1]
    assert o1lo.tolist() == [-15, -20, 0] and o1hi.tolist() == [30, 24, 1]  # 20 m pad in the projection plane only
    assert o2lo.tolist() == [-20, -20, 0] and o2hi.tolist() == [25, 24, 1]
    # projection along x: the plane is (y, z); z is longer here
    (t1lo, t1hi, _, _), _ = T._split_boxes(np.array([0, 0, 0], np.float32), np.array([1, 2, 8], np.float32), 0)
    assert t1lo.tolist() == [0, 0, 4] and t1hi.tolist() == [1, 2, 8]
    c = T._Cloud(np.array([[5, 1, 0.5], [4.999, 1, 0.5], [5.001, 1, 0.5]], np.float32))
    assert len(T._crop(c, [5, 0, 0], [10, 4, 1])) == 2 and len(T._crop(c, [0, 0, 0], [5, 4, 1])) == 2  # inclusive both ways


def test_tile_point_clouds_structure(tmp_path):
    rng = np.random.default_rng(5)
    a = _surface(rng, 6000, 0, 100, 0, 60)
    b = _surface(rng, 5000, 10, 120, -5, 55)                     # overlap box: x 10..100, y 0..55
    rgb = rng.integers(0, 256, (len(a), 3)).astype(np.uint8)
    pa, pb, out = str(tmp_path / "a.ply"), str(tmp_path / "b.ply"), str(tmp_path / "tiles")
    _write_cloud(pa, a, rgb)
    _write_cloud(pb, b)
    assert T.tile_point_clouds(pa, pb, 1000, 0, False, 0.0, 0.0, -1, out, False) is True
    assert T.tile_point_clouds(str(tmp_path / "missing.ply"), pb, 1000, 0, False, 0.0, 0.0, -1, out, False) is False
    names = sorted(os.listdir(os.path.join(out, "non_overlap")))
    n_tiles = len(names) // 2
    assert n_tiles >= 5 and names == sorted([f"{k}_tile_{i}.ply" for k in ("source", "target") for i in range(n_tiles)])
    lo = np.maximum(a.min(0), b.min(0)); hi = np.minimum(a.max(0), b.max(0))
    inside = lambda p: p[np.all((p >= lo) & (p <= hi), axis=1)]
    for kind, cloud in (("source", inside(a)), ("target", inside(b))):
        seen = []
        for i in range(n_tiles):
            t, f = read_ply(os.path.join(out, "non_overlap", f"{kind}_tile_{i}.ply"))
            o, _ = read_ply(os.path.join(out, "overlap", f"{kind}_tile_{i}_overlap.ply"))
            assert len(t) < 1000 and len(t) > 1
            assert (kind == "source") == ("red" in f)                       # colours travel with the cloud that has them
            tset = {tuple(r) for r in t.astype(np.float32)}
            assert tset <= {tuple(r) for r in o.astype(np.float32)}        # the overlap twin contains its tile
            # ... and everything within 20 m of the tile's box in x and y
            tl, th = t.min(0), t.max(0)
            near = cloud[(cloud[:, 0] >= tl[0] - 19) & (cloud[:, 0] <= th[0] + 19) & (cloud[:, 1] >= tl[1] - 19) & (cloud[:, 1] <= th[1] + 19)]
            assert {tuple(r) for r in near} <= {tuple(r) for r in o.astype(np.float32)}
            seen.append(t.astype(np.float32))
        allp = np.concatenate(seen)
        # the tiles cover the cropped cloud; only points exactly on a cut may appear twice
        assert {tuple(r) for r in allp} == {tuple(r) for r in cloud}
        assert len(allp) - len(cloud) <= 3
    # tile 0 is the upper-most half of every split (the reference recurses into the upper half first)
    t0, _ = read_ply(os.path.join(out, "non_overlap", "source_tile_0.ply"))
    assert t0[:, 0].max() == inside(a)[:, 0].max()


def test_resave_point_cloud(tmp_path):
    rng = np.random.default_rng(6)
    a = _surface(rng, 50, 0, 1, 0, 1)
    p1, p2 = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    with open(p1, "w") as f:  # ascii in, binary out
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(f, a, fmt="%.9g")
    _write_cloud(p2, a)
    assert T.resave_point_cloud(p1, p2, False) is True
    assert b"binary_little_endian" in open(p1, "rb").read(64)
    assert np.array_equal(read_ply(p1)[0].astype(np.float32), a)


@pytest.mark.gpu
def test_voxel_grid_pcl_layout_vs_numpy_and_full_tiling(tmp_path):
    import torch
    from fusion4landslide_amd import engine
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    a = _surface(rng, 40000, 2600000, 2600060, 1200000, 1200040)   # georeferenced coordinates: float32 cell arithmetic matters
    a[:, 2] += 1500
    for leaf in (0.5, 0.13, 7.0):
        pts, cnt, vop = engine.voxel_downsample(torch.from_numpy(a).cuda(), leaf, return_map=True, layout="pcl")
        rp, rc, rv = O.voxel_grid_pcl(a, leaf)
        assert len(rp) == pts.shape[0] and np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(vop.cpu().numpy(), rv)
        assert np.abs(pts.cpu().numpy() - rp).max() <= 1e-6
    b = _surface(rng, 30000, 2600010, 2600070, 1199995, 1200035)
    b[:, 2] += 1500
    pa, pb, out = str(tmp_path / "a.ply"), str(tmp_path / "b.ply"), str(tmp_path / "tiles")
    _write_cloud(pa, a, rng.integers(0, 256, (len(a), 3)).astype(np.uint8))
    _write_cloud(pb, b)
    assert T.tile_point_clouds(pa, pb, 4000, 0, True, 0.0, 0.0, -1, out, False) is True   # leaf from the median spacing
    n_tiles = len(os.listdir(os.path.join(out, "non_overlap"))) // 2
    assert n_tiles >= 2
    total = 0
    for i in range(n_tiles):
        t, f = read_ply(os.path.join(out, "non_overlap", f"source_tile_{i}.ply"))
        assert 1 < len(t) < 4000 and "red" in f
        total += len(t)
    # thinned: fewer points than went in, none lost to the recursion
    lo = np.maximum(a.min(0), b.min(0)); hi = np.minimum(a.max(0), b.max(0))
    n_in = int(np.all((a >= lo) & (a <= hi), axis=1).sum())
    assert total < n_in
