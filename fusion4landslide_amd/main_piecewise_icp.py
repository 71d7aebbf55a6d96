"""Counterpart of the reference's main_piecewise_icp.py: `--config` (flat yaml) -> cfg -> per tile `Piecewise_ICP(cfg)`.

Same cfg keys and output files (main_piecewise_icp.py:20-102).  Like the reference (:64-80), an empty
`<output_root>/tiled_data/` is filled by `point_cloud_tiling` (src/functions.py:147-177 -> the tiler mirror
cpp_core/pcd_tiling/build/pcd_tiling.py); tiles already there are used as they are.

    python -m fusion4landslide_amd.main_piecewise_icp --config configs/landslide/piecewise_icp_brienz.yaml [--engine patch_icp]
"""
import argparse
import copy
import glob
import os
import os.path as osp
import re
import time

from .src.functions import point_cloud_tiling
from .src.piecewise_icp import Piecewise_ICP
from .utils.common import AttrDict, access_device, dir_exist, get_logger, load_yaml


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='./configs/landslide/piecewise_icp_brienz.yaml',
                        help='Path to config file.')
    parser.add_argument('--engine', type=str, default=None, choices=['reference_octree', 'patch_icp'])
    args = parser.parse_args(argv)
    cfg = load_yaml(args.config, keep_sub_directory=False)
    cfg['output_root'] = osp.join(cfg['output_dir'], cfg['output_folder'])
    log_dir = osp.join(cfg['output_root'], 'logs')
    os.makedirs(log_dir, exist_ok=True)
    log_save_path = osp.join(log_dir, 'piecewise_icp_{}.log'.format(time.strftime('%Y%m%d_%H%M%S')))
    cfg['logging'] = get_logger(log_save_path)
    cfg = AttrDict(cfg)
    if args.engine:
        cfg.engine = args.engine
    cfg.device = access_device()
    start_time = time.time()

    cfg.tile_dir = osp.join(cfg.output_root, 'tiled_data')
    dir_exist(cfg.tile_dir)
    if not any(os.listdir(cfg.tile_dir)):  # main_piecewise_icp.py:66-78
        config = copy.copy(cfg)
        config.data_dir = cfg.input_root
        config.src_name = cfg.src_pcd
        config.tgt_name = cfg.tgt_pcd
        point_cloud_tiling(config)
    else:
        cfg.logging.info('Skip point cloud tiling. Tiles will be loaded from %s.', cfg.tile_dir)
    src_tiles = sorted(glob.glob(osp.join(cfg.tile_dir, 'overlap', "source_tile_*")),
                       key=lambda x: int(re.search(r'\d+', osp.basename(x)).group()))
    cfg.logging.info(f'Num. of tile(s) from source/target point cloud: {len(src_tiles)}')
    for tile_i, src_path in enumerate(src_tiles):
        tgt_path = src_path.replace('source_tile_', 'target_tile_')
        assert osp.exists(tgt_path)
        cfg.tile_id = re.findall(r'\d+', osp.basename(src_path))[0]
        cfg.src_tile_overlap_path = src_path
        cfg.tgt_tile_overlap_path = tgt_path
        Piecewise_ICP(cfg)
    cfg.logging.info(f"Displacement estimation is done! Save results to: '{cfg.output_root}'. "
                     f"Total time taken: {time.time() - start_time:.1f} seconds.")


if __name__ == '__main__':
    main()
