"""The tiling front end: cpp_core/pcd_tiling/pcd_tiling.cpp:709-871 -> f4l_tile_point_clouds (fusion4landslide_amd/csrc/tiling.hip) behind
the shim fusion4landslide_amd/cpp_core/pcd_tiling/build/pcd_tiling.py.  PCL is not available here, so the checker is a numpy
restatement of the reference's steps (oracle/pcd_tiling_ref.py; parity unpinned): CPU tests pin the restatement's structure, GPU tests
compare the library entry with it FILE FOR FILE, byte for byte, on the reference's ten-argument call."""
import os

import numpy as np
import pytest

from fusion4landslide_amd.utils.ply import read_ply
from oracle import pcd_tiling_ref as R


def _write_cloud(path, xyz, rgb=None):
    R._write(path, R._Cloud(xyz, rgb))


def _surface(rng, n, x0, x1, y0, y1):
    xy = np.c_[rng.uniform(x0, x1, n), rng.uniform(y0, y1, n)]
    return np.c_[xy, 0.5 * np.sin(0.05 * xy[:, 0]) * np.cos(0.04 * xy[:, 1])].astype(np.float32)


def test_split_boxes_halve_the_longer_in_plane_side():
    lo, hi = np.array([0, 0, 0], np.float32), np.array([10, 4, 1], np.float32)
    (t1lo, t1hi, o1lo, o1hi), (t2lo, t2hi, o2lo, o2hi) = R._split_boxes(lo, hi, 2)
    assert t1lo.tolist() == [5, 0, 0] and t1hi.tolist() == [10, 4, 1]      # upper half first
    assert t2lo.tolist() == [0, 0, 0] and t2hi.tolist() == [5, 4, 1]
    assert o1lo.tolist() == [-15, -20, 0] and o1hi.tolist() == [30, 24, 1]  # 20 m pad in the projection plane only
    assert o2lo.tolist() == [-20, -20, 0] and o2hi.tolist() == [25, 24, 1]
    # projection along x: the plane is (y, z); z is longer here
    (t1lo, t1hi, _, _), _ = R._split_boxes(np.array([0, 0, 0], np.float32), np.array([1, 2, 8], np.float32), 0)
    assert t1lo.tolist() == [0, 0, 4] and t1hi.tolist() == [1, 2, 8]
    c = R._Cloud(np.array([[5, 1, 0.5], [4.999, 1, 0.5], [5.001, 1, 0.5]], np.float32))
    assert len(R._crop(c, [5, 0, 0], [10, 4, 1])) == 2 and len(R._crop(c, [0, 0, 0], [5, 4, 1])) == 2  # inclusive both ways


def test_restatement_structure(tmp_path):
    """The checker itself (host only): every tile below the limit, the overlap twin contains its tile and everything within 20 m of it
    in the projection plane, the tiles cover the cropped cloud, tile 0 is the upper-most half of every split."""
    rng = np.random.default_rng(5)
    a = _surface(rng, 6000, 0, 100, 0, 60)
    b = _surface(rng, 5000, 10, 120, -5, 55)                     # overlap box: x 10..100, y 0..55
    rgb = rng.integers(0, 256, (len(a), 3)).astype(np.uint8)
    pa, pb, out = str(tmp_path / "a.ply"), str(tmp_path / "b.ply"), str(tmp_path / "tiles")
    _write_cloud(pa, a, rgb)
    _write_cloud(pb, b)
    assert R.tile_point_clouds(pa, pb, 1000, 0, False, 0.0, 0.0, -1, out, False) is True
    assert R.tile_point_clouds(str(tmp_path / "missing.ply"), pb, 1000, 0, False, 0.0, 0.0, -1, out, False) is False
    _check_structure(out, a, b, 1000)


def _check_structure(out, a, b, limit):
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
            assert len(t) < limit and len(t) > 1
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


def _same_files(dir_a, dir_b):
    for sub in ("non_overlap", "overlap"):
        na, nb = sorted(os.listdir(os.path.join(dir_a, sub))), sorted(os.listdir(os.path.join(dir_b, sub)))
        assert na == nb and len(na) >= 2, (sub, na, nb)
        for name in na:
            assert open(os.path.join(dir_a, sub, name), "rb").read() == open(os.path.join(dir_b, sub, name), "rb").read(), (sub, name)
    return len(os.listdir(os.path.join(dir_a, "non_overlap"))) // 2


@pytest.mark.gpu
def test_library_tiler_equals_the_restatement_file_for_file(tmp_path):
    """f4l_tile_point_clouds through the drop-in module, on the reference's ten-argument call, against oracle/pcd_tiling_ref.py: the same
    files with the same bytes -- without and with the voxel grid (given leaf, and leaf = median spacing), colours on one epoch, every
    projection direction, georeferenced coordinates (float32 cell arithmetic), points exactly on a cut, ascii and double-precision
    inputs; the missing-file and the too-many-coincident-points answers; `resave_point_cloud`."""
    from fusion4landslide_amd.cpp_core.pcd_tiling.build import pcd_tiling as T
    rng = np.random.default_rng(5)
    a = _surface(rng, 6000, 0, 100, 0, 60)
    b = _surface(rng, 5000, 10, 120, -5, 55)
    a[:40, 0] = 55.0                                               # points exactly on the first cut of the overlap box (x 10..100)
    rgb = rng.integers(0, 256, (len(a), 3)).astype(np.uint8)
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    _write_cloud(pa, a, rgb)
    _write_cloud(pb, b)
    for tag, args in (("plain", (1000, 0, False, 0.0, 0.0, -1)), ("dir0", (1500, 100, False, 0.05, 0.0, 0)), ("dir1", (1500, 100, False, 0.05, 0.0, 1)),
                      ("dir2", (800, 100, False, 0.05, 5.0, 2)), ("grid", (700, 0, True, 1.5, 0.0, -1)), ("grid_auto", (900, 0, True, 0.0, 0.0, -1))):
        got, ref = str(tmp_path / f"got_{tag}"), str(tmp_path / f"ref_{tag}")
        assert T.tile_point_clouds(pa, pb, *args, got, False) is True
        assert R.tile_point_clouds(pa, pb, *args, ref, False) is True
        n_tiles = _same_files(got, ref)
        assert n_tiles >= 2, tag
    _check_structure(str(tmp_path / "got_plain"), a, b, 1000)
    assert T.tile_point_clouds(str(tmp_path / "missing.ply"), pb, 1000, 0, False, 0.0, 0.0, -1, str(tmp_path / "none"), False) is False
    # georeferenced clouds, colours, the voxel grid from the median spacing (the configuration main_fusion.py:113-123 asks for)
    g1 = _surface(rng, 40000, 2600000, 2600060, 1200000, 1200040)
    g1[:, 2] += 1500
    g2 = _surface(rng, 30000, 2600010, 2600070, 1199995, 1200035)
    g2[:, 2] += 1500
    p1, p2 = str(tmp_path / "g1.ply"), str(tmp_path / "g2.ply")
    _write_cloud(p1, g1, rng.integers(0, 256, (len(g1), 3)).astype(np.uint8))
    _write_cloud(p2, g2)
    got, ref = str(tmp_path / "got_geo"), str(tmp_path / "ref_geo")
    assert T.tile_point_clouds(p1, p2, 4000, 0, True, 0.0, 0.0, -1, got, False) is True
    assert R.tile_point_clouds(p1, p2, 4000, 0, True, 0.0, 0.0, -1, ref, False) is True
    n_tiles = _same_files(got, ref)
    total = sum(len(read_ply(os.path.join(got, "non_overlap", f"source_tile_{i}.ply"))[0]) for i in range(n_tiles))
    lo = np.maximum(g1.min(0), g2.min(0)); hi = np.minimum(g1.max(0), g2.max(0))
    assert n_tiles >= 2 and total < int(np.all((g1 >= lo) & (g1 <= hi), axis=1).sum())  # thinned
    # an ascii PLY and a double-precision one: the same tiles as their float32 binary twins
    pa_ascii, pb_double = str(tmp_path / "a_ascii.ply"), str(tmp_path / "b_double.ply")
    with open(pa_ascii, "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(a)}\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                "property uchar green\nproperty uchar blue\nend_header\n")
        for p, c in zip(a, rgb):
            f.write("%.9g %.9g %.9g %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]))
    from fusion4landslide_amd.utils.ply import write_ply
    write_ply(pb_double, b, dtype="float64")
    got2 = str(tmp_path / "got_ascii")
    assert T.tile_point_clouds(pa_ascii, pb_double, 1000, 0, False, 0.0, 0.0, -1, got2, False) is True
    _same_files(got2, str(tmp_path / "ref_plain"))
    # more than maxPointsPerTile coincident points: refused (the reference recurses until its stack overflows)
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (50, 1))
    ps = str(tmp_path / "same.ply")
    _write_cloud(ps, same)
    with pytest.raises(ValueError):
        T.tile_point_clouds(ps, ps, 10, 0, False, 0.0, 0.0, -1, str(tmp_path / "stuck"), False)
    # resave_point_cloud: ascii in, binary out, same coordinates
    assert T.resave_point_cloud(pa_ascii, pb_double, False) is True
    assert b"binary_little_endian" in open(pa_ascii, "rb").read(64)
    assert np.array_equal(read_ply(pa_ascii)[0].astype(np.float32), a) and np.array_equal(read_ply(pb_double)[0].astype(np.float32), b)
    assert T.resave_point_cloud(str(tmp_path / "missing.ply"), pb, False) is False


def test_resave_point_cloud_of_the_restatement(tmp_path):
    rng = np.random.default_rng(6)
    a = _surface(rng, 50, 0, 1, 0, 1)
    p1, p2 = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    with open(p1, "w") as f:  # ascii in, binary out
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(f, a, fmt="%.9g")
    _write_cloud(p2, a)
    assert R.resave_point_cloud(p1, p2, False) is True
    assert b"binary_little_endian" in open(p1, "rb").read(64)
    assert np.array_equal(read_ply(p1)[0].astype(np.float32), a)


@pytest.mark.gpu
def test_voxel_grid_pcl_layout_vs_numpy():
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
