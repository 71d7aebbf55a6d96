"""The fusion-path entry (fusion4landslide_amd/main_fusion.py + src/coarse_to_fine_matching.py, the counterparts of the
reference's main_fusion.py:134-148 and Coarse2Fine.implement_c2f_matching) and its result writer `save_process_dvf`
(src/coarse_to_fine_matching_base.py:3459-3600).  The writer is host code (CPU test); the entry runs on the GPU."""
import os

import numpy as np
import pytest


def test_save_process_dvf_writes_the_reference_files(tmp_path):
    """File names, column layout, '%.6f', and the two planted rows of the *_visualize files (base:3477-3516): the files read back to
    the rows they were given."""
    from fusion4landslide_amd.src.coarse_to_fine_matching import save_process_dvf
    rng = np.random.default_rng(0)
    src = rng.uniform(0, 50, (400, 3))
    dense = np.c_[src, src + rng.normal(0, 0.05, (400, 3))].astype(np.float32)
    sparse = dense[::3]
    files = save_process_dvf(str(tmp_path), "7", "brienz_tls", dense, sparse)
    names = sorted(os.path.basename(f) for f in files)
    assert names == ["c2f_dense_dvfms_src2tgt_tile_7.txt", "c2f_dense_dvfms_src2tgt_visualize_tile_7.txt",
                     "c2f_dense_dvfs_src2tgt_tile_7.txt", "c2f_sparse_dvfms_src2tgt_visualize_tile_7.txt"]
    res = tmp_path / "results"
    rows = np.loadtxt(res / "c2f_dense_dvfs_src2tgt_tile_7.txt")
    assert rows.shape == (400, 6) and np.abs(rows - dense).max() <= 5.1e-7  # '%.6f'
    first = open(res / "c2f_dense_dvfs_src2tgt_tile_7.txt").readline().split()
    assert len(first) == 6 and all(len(v.split(".")[1]) == 6 for v in first)
    mag = np.linalg.norm(dense[:, 3:] - dense[:, :3], axis=1)
    dvfms = np.loadtxt(res / "c2f_dense_dvfms_src2tgt_tile_7.txt")
    assert dvfms.shape == (400, 4) and np.abs(dvfms[:, :3] - dense[:, :3]).max() <= 5.1e-7 and np.abs(dvfms[:, 3] - mag).max() <= 1e-6
    vis = np.loadtxt(res / "c2f_dense_dvfms_src2tgt_visualize_tile_7.txt")
    assert vis[0, 3] == 0 and vis[1, 3] == 5 and np.abs(vis[2:, 3] - mag[2:]).max() <= 1e-6  # max_mag_visualize of brienz_tls (:3482-3483)
    svis = np.loadtxt(res / "c2f_sparse_dvfms_src2tgt_visualize_tile_7.txt")
    assert svis.shape == (len(sparse), 4) and svis[0, 3] == 0 and svis[1, 3] == 5
    # the other data sets' colour scale, and the single-case file names (:3570-3600)
    files = save_process_dvf(str(tmp_path / "rock"), "0", "rockfall_simulator", dense, sparse, multiple_case=False)
    assert sorted(os.path.basename(f) for f in files) == ["c2f_dvfms_src2tgt.txt", "c2f_dvfms_src2tgt_discrete_visualize_0_5.txt",
                                                          "c2f_dvfms_src2tgt_visualize_0_5.txt", "c2f_dvfs_src2tgt.txt"]
    assert np.loadtxt(tmp_path / "rock" / "results" / "c2f_dvfms_src2tgt_visualize_0_5.txt")[1, 3] == pytest.approx(0.06)


def test_row_writer_is_numpy_savetxt_byte_for_byte(tmp_path):
    """f4l_write_rows_txt (what save_process_dvf writes its tables with) against `np.savetxt(path, rows, delimiter=" ", fmt="%.6f")`,
    the reference's call (src/coarse_to_fine_matching_base.py:3477-3537): every byte, on random rows and on the values where a
    six-decimal writer can go wrong -- exact ties of the seventh decimal (odd multiples of 1/128: to even, like printf), signed
    zeros and tiny negatives ('-0.000000'), the float32 extremes, 10^9 and beyond, infinities, NaN."""
    import ctypes as C
    from fusion4landslide_amd._lib import check, lib
    rng = np.random.default_rng(3)
    edge = np.array([0.0, -0.0, 1 / 128, -1 / 128, 3 / 128, 1e-7, -1e-7, 5e-7, -5e-7, 4.9999997e-7, 1e9, -1e9, 999999999.0, 1.5e9, 3.4e38,
                     -3.4e38, np.inf, -np.inf, np.nan, 0.9999995, 0.99999994, 123456.789, 2.5e-6, 1.5e-6, 0.5, 1e-45, -1e-45, 65504.0,
                     16777216.0, 0.1, 0.2, 0.3], np.float32)
    edge = np.concatenate([edge, (np.arange(1, 4000, 2) / 128.0).astype(np.float32), -(np.arange(1, 4000, 2) / 128.0).astype(np.float32)])
    for ncols in (6, 4, 1):
        rows = np.concatenate([np.resize(edge, (len(edge) // ncols * ncols,)).reshape(-1, ncols),
                               (rng.normal(size=(20_000, ncols)) * 10.0 ** rng.integers(-7, 6, (20_000, 1))).astype(np.float32)])
        ref, out = tmp_path / f"ref_{ncols}.txt", tmp_path / f"out_{ncols}.txt"
        np.savetxt(ref, rows, delimiter=" ", fmt="%.6f")
        check(lib().f4l_write_rows_txt(str(out).encode(), rows.ctypes.data_as(C.c_void_p), rows.shape[0], ncols), "f4l_write_rows_txt")
        assert open(ref, "rb").read() == open(out, "rb").read(), ncols
    check(lib().f4l_write_rows_txt(str(tmp_path / "empty.txt").encode(), None, 0, 6), "f4l_write_rows_txt")
    assert open(tmp_path / "empty.txt", "rb").read() == b""
    assert lib().f4l_write_rows_txt(str(tmp_path / "no_such_dir" / "x.txt").encode(), rows.ctypes.data_as(C.c_void_p), 1, 1) != 0


@pytest.mark.gpu
def test_main_fusion_entry_runs_the_tiles_and_writes_the_dvf_files(tmp_path, monkeypatch):
    """`python -m fusion4landslide_amd.main_fusion --config <yaml>` on a synthetic two-tile data set already tiled (the tiler
    is skipped like in the reference when tiled_data/ is not empty): the reference's nested yaml keys, both tiles visited in
    numeric order, the c2f_* files of save_process_dvf per tile; the dense rows are [s, T s] of the registered patches and
    recover the planted motion of the stable blocks.  Then the fusion branch through `run()` with a second set of point matches
    in the role of the lifted 2D matches."""
    import torch
    import yaml
    from fusion4landslide_amd import main_fusion, synthetic
    from fusion4landslide_amd.utils.ply import write_ply
    out_root = tmp_path / "out" / "demo_run"
    tiles = out_root / "tiled_data" / "overlap"
    os.makedirs(tiles)
    clouds = {}
    for t, seed in ((0, 0), (1, 5)):
        c = synthetic.two_epoch_cloud(60_000, 9, 1.386, seed=seed, roughness=0.05)
        write_ply(str(tiles / f"source_tile_{t}_overlap.ply"), c["src"])
        write_ply(str(tiles / f"target_tile_{t}_overlap.ply"), c["tgt"])
        clouds[t] = c
    cfg = dict(
        misc=dict(verbose=True, save_interim=False),
        path_name=dict(input_root=str(tmp_path), output_dir=str(tmp_path / "out"), output_folder="demo_run"),
        data=dict(dataset="brienz_tls", src_pcd="a.ply", tgt_pcd="b.ply", multiple_case=True),
        method=dict(tiling_type="xy_tiling", max_pts_per_tile=1000000, min_pts_per_tile=5000, voxel_size_init=0.1, partition=True,
                    partition_type="supervoxel", fine_matching_fusion=False, fine_matching_only_3d=True, fine_matching_only_2d=False,
                    remove_low_quality_patch_matches=True, num_min_matches_for_quality_check=10, thres_dist_diff=0.5, thres_inlier_ratio=0.15,
                    num_min_fine_match=10, weighting_svd=False, icp_refine=True, output_tgt2src=False, assign_type="assign_then_nn"),
        parameter_setting=dict(n_normals=30, icp_threshold=0.1, max_magnitude=5))
    path = tmp_path / "fusion_3d.yaml"
    yaml.safe_dump(cfg, open(path, "w"))
    from fusion4landslide_amd.cpp_core.supervoxel_segmentation.build import supervoxel
    seen = []
    run_before = main_fusion.run
    main_fusion.run = lambda *a, **k: (seen.append((supervoxel.SEGMENTATION, a[2] if len(a) > 2 else k.get("tiles_per_launch"))), run_before(*a, **k))[1]
    calls = []
    compute_before = supervoxel.computeSupervoxelDevice

    def recorder(xyz_dev, k, r, out="None", **kw):
        lab, K = compute_before(xyz_dev, k, r, out, **kw)
        calls.append((xyz_dev.cpu().numpy(), k, r, out, lab.cpu().numpy(), K))
        return lab, K
    monkeypatch.setattr(supervoxel, "computeSupervoxelDevice", recorder)
    try:
        # the DEFAULT mode, un-stubbed: no --partition = the reference's labels, no --tiles-per-launch = tile by tile
        main_fusion.main(["--config", str(path)])
        monkeypatch.setattr(supervoxel, "computeSupervoxelDevice", compute_before)
        monkeypatch.setenv("F4L_SV_MODE", "fast")
        with pytest.raises(SystemExit):            # an invalid mode is refused before anything runs
            main_fusion.main(["--config", str(path)])
        monkeypatch.delenv("F4L_SV_MODE")
    finally:
        main_fusion.run = run_before
    assert seen == [("identical", 1)] and supervoxel.SEGMENTATION == "identical"
    res = out_root / "results"
    from scipy.spatial import cKDTree
    from fusion4landslide_amd import engine
    for t in (0, 1):
        dvfs = np.loadtxt(res / f"c2f_dense_dvfs_src2tgt_tile_{t}.txt")
        dvfms = np.loadtxt(res / f"c2f_dense_dvfms_src2tgt_tile_{t}.txt")
        vis = np.loadtxt(res / f"c2f_dense_dvfms_src2tgt_visualize_tile_{t}.txt")
        sparse = np.loadtxt(res / f"c2f_sparse_dvfms_src2tgt_visualize_tile_{t}.txt")
        assert dvfs.shape[1] == 6 and dvfms.shape == (len(dvfs), 4) and vis.shape == dvfms.shape and sparse.shape[1] == 4
        assert 30_000 < len(dvfs) <= 60_000 and vis[0, 3] == 0 and vis[1, 3] == 5
        assert np.allclose(dvfms[:, 3], np.linalg.norm(dvfs[:, 3:] - dvfs[:, :3], axis=1), atol=2e-6)
        # every row starts at a point of the tile's source epoch; the field of the stable blocks is a few centimetres
        assert cKDTree(clouds[t]["src"].astype(np.float64)).query(dvfs[:, :3], k=1)[0].max() < 2e-6  # ('%.6f' of float32 coordinates)
        assert np.median(dvfms[:, 3]) < 0.12
    # the partition files of the default mode carry the REFERENCE's labels: column 6 of each is what the one-core replay of
    # supervoxel_segmentation.h:117-237 (csrc/supervoxel_host.cpp, pinned by the reference-compiled fixtures) gives for the same call
    assert len(calls) == 4 and all(c[1] == 30 for c in calls)
    monkeypatch.setenv("F4L_SV_EXACT_HOST", "1")
    for (xyz, k, r, out, lab, K), cloud in zip(calls, (clouds[0]["src"], clouds[0]["tgt"], clouds[1]["src"], clouds[1]["tgt"])):
        assert np.array_equal(xyz, cloud)              # (the tile's PLY as it was written, read once)
        replay, K_r = supervoxel.computeSupervoxelArray(xyz, k, r)
        assert np.array_equal(replay, lab) and lab.max() + 1 == K == K_r
        # the partition file (written by a writer thread; the run has drained them) = what the drop-in `computeSupervoxel` writes
        part = np.loadtxt(out)
        assert part.shape == (len(lab), 7) and np.array_equal(part[:, 6].astype(np.int64), lab)
        assert np.array_equal(part[:, :3].astype(np.float32), xyz)
    monkeypatch.delenv("F4L_SV_EXACT_HOST")
    # ... byte for byte: the serial order of work (F4L_ASYNC_IO=0) and the module's file-in / file-out function give the same files
    monkeypatch.setenv("F4L_ASYNC_IO", "0")
    res_serial = None
    try:
        cfg_s = dict(cfg, path_name=dict(cfg["path_name"], output_folder="serial_run"))
        os.makedirs(tmp_path / "out" / "serial_run" / "tiled_data" / "overlap")
        for name in os.listdir(tiles):
            os.link(tiles / name, tmp_path / "out" / "serial_run" / "tiled_data" / "overlap" / name)
        yaml.safe_dump(cfg_s, open(tmp_path / "fusion_3d_serial.yaml", "w"))
        main_fusion.main(["--config", str(tmp_path / "fusion_3d_serial.yaml")])
        res_serial = tmp_path / "out" / "serial_run"
    finally:
        monkeypatch.delenv("F4L_ASYNC_IO")
    for sub in ("results", "supervoxel_partition"):
        names = sorted(os.listdir(out_root / sub))
        assert names == sorted(os.listdir(res_serial / sub)) and len(names) >= 4
        for name in names:
            assert open(out_root / sub / name, "rb").read() == open(res_serial / sub / name, "rb").read(), name
    direct = tmp_path / "direct_partition.txt"
    lab_direct = supervoxel.computeSupervoxel(str(tiles / "source_tile_0_overlap.ply"), 30, calls[0][2], str(direct))
    assert np.array_equal(lab_direct, calls[0][4])
    assert open(direct, "rb").read() == open(out_root / "supervoxel_partition" / "partition_of_input_src_tile_0.txt", "rb").read()
    # the same tiles with the all-device segmentation (opt-in) and 8 tiles around one per-patch launch: every file again; and for
    # one partition the batching moves a row by at most the last printed digit (1e-9 m in a transform, whatever the batch)
    def run_into(folder, argv):
        cfgx = dict(cfg, path_name=dict(cfg["path_name"], output_folder=folder))
        os.makedirs(tmp_path / "out" / folder / "tiled_data" / "overlap")
        for name in os.listdir(tiles):
            os.link(tiles / name, tmp_path / "out" / folder / "tiled_data" / "overlap" / name)
        px = tmp_path / f"fusion_3d_{folder}.yaml"
        yaml.safe_dump(cfgx, open(px, "w"))
        main_fusion.main(["--config", str(px)] + argv)
        return tmp_path / "out" / folder / "results"
    res_p1 = run_into("parallel_one_by_one", ["--partition", "parallel"])
    res_p8 = run_into("parallel_batched", ["--partition", "parallel", "--tiles-per-launch", "8"])
    for t in (0, 1):
        for name in (f"c2f_dense_dvfs_src2tgt_tile_{t}.txt", f"c2f_sparse_dvfms_src2tgt_visualize_tile_{t}.txt"):
            a, b = np.loadtxt(res_p1 / name), np.loadtxt(res_p8 / name)
            assert a.shape == b.shape and np.abs(a - b).max() <= 1.01e-6, name
        assert os.path.exists(tmp_path / "out" / "parallel_batched" / "supervoxel_partition" / f"partition_of_input_src_tile_{t}.txt")
    # the fusion branch: the same tile with lifted "2D" matches attached to the cfg
    cfg2, _ = main_fusion.build_config(str(path))
    cfg2.method.fine_matching_only_3d, cfg2.method.fine_matching_fusion, cfg2.method.weighting_svd = False, True, True
    cfg2.path_name.output_root = str(tmp_path / "out" / "fusion_run")
    os.makedirs(tmp_path / "out" / "fusion_run" / "tiled_data" / "overlap")
    for name in ("source_tile_0_overlap.ply", "target_tile_0_overlap.ply"):
        os.link(tiles / name, tmp_path / "out" / "fusion_run" / "tiled_data" / "overlap" / name)
    c = clouds[0]
    d, j = cKDTree(c["tgt"].astype(np.float64)).query(c["src"].astype(np.float64), k=2, distance_upper_bound=0.3)
    cfg2.point_matches_from_2d = np.where(np.isfinite(d[:, 1]) & (np.arange(len(d)) % 3 == 0), j[:, 1], -1)
    supervoxel.SEGMENTATION = "parallel"
    try:
        main_fusion.run(cfg2)
    finally:
        supervoxel.SEGMENTATION = "identical"
    fus = np.loadtxt(tmp_path / "out" / "fusion_run" / "results" / "c2f_dense_dvfs_src2tgt_tile_0.txt")
    assert fus.shape[1] == 6 and len(fus) > 30_000 and np.median(np.linalg.norm(fus[:, 3:] - fus[:, :3], axis=1)) < 0.12
