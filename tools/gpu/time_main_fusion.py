"""Wall time of the fusion entry per tile (python -m fusion4landslide_amd.main_fusion) on synthetic 1 M-point tiles already tiled:
read PLY -> partition (the reference's labels, partition files written) -> matches -> patch loop -> result files.
Usage: time_main_fusion.py [n_points_per_tile] [tiles]"""
import os, sys, time, tempfile, cProfile, pstats
import numpy as np, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import main_fusion, synthetic
from fusion4landslide_amd.utils.ply import write_ply

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
root = tempfile.mkdtemp(prefix="f4l_mf_")
tiles = os.path.join(root, "out", "run", "tiled_data", "overlap")
os.makedirs(tiles)
for t in range(T):
    c = synthetic.two_epoch_cloud(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, seed=t, roughness=0.05)
    write_ply(os.path.join(tiles, f"source_tile_{t}_overlap.ply"), c["src"])
    write_ply(os.path.join(tiles, f"target_tile_{t}_overlap.ply"), c["tgt"])
cfg = dict(misc=dict(verbose=False, save_interim=False),
           path_name=dict(input_root=root, output_dir=os.path.join(root, "out"), output_folder="run"),
           data=dict(dataset="brienz_tls", src_pcd="a.ply", tgt_pcd="b.ply", multiple_case=True),
           method=dict(tiling_type="xy_tiling", max_pts_per_tile=1000000, min_pts_per_tile=5000, voxel_size_init=0.1, partition=True,
                       partition_type="supervoxel", fine_matching_fusion=False, fine_matching_only_3d=True, fine_matching_only_2d=False,
                       remove_low_quality_patch_matches=True, num_min_matches_for_quality_check=10, thres_dist_diff=0.5, thres_inlier_ratio=0.15,
                       num_min_fine_match=10, weighting_svd=False, icp_refine=True, output_tgt2src=False, assign_type="assign_then_nn"),
           parameter_setting=dict(n_normals=30, icp_threshold=0.1, max_magnitude=5))
path = os.path.join(root, "cfg.yaml")
yaml.safe_dump(cfg, open(path, "w"))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
main_fusion.main(["--config", path])
pr.disable()
wall = time.perf_counter() - t0
print(f"main_fusion: {T} tiles of {n} points: {wall:.2f} s wall = {wall / T:.2f} s per tile")
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io; s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print("\n".join(l[:150] for l in s.getvalue().split("\n")[6:40]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print("\n".join(l[:150] for l in s.getvalue().split("\n")[6:34]))
