"""Counterpart of the reference's main_piecewise_icp.py: `--config` (flat yaml) -> cfg -> per tile `Piecewise_ICP(cfg)`.

Same cfg keys and output files (main_piecewise_icp.py:20-102).  Like the reference (:64-80), an empty
`<output_root>/tiled_data/` is filled by `point_cloud_tiling` (src/functions.py:147-177 -> the tiler mirror
cpp_core/pcd_tiling/build/pcd_tiling.py); tiles already there are used as they are.

    python -m fusion4landslide_amd.main_piecewise_icp --config configs/landslide/piecewise_icp_brienz.yaml [--engine patch_icp]
"""
import argparse
import copy
import os
import os.path as osp
import time

from .src.piecewise_icp import Piecewise_ICP
from .utils.common import AttrDict, access_device, get_logger, load_yaml
from .utils.tiles import for_each_tile, prepare_tiles


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='./configs/landslide/piecewise_icp_brienz.yaml', help='Path to config file.')
    parser.add_argument('--engine', type=str, default=None, choices=['reference_octree', 'patch_icp'])
    args = parser.parse_args(argv)
    cfg = load_yaml(args.config, keep_sub_directory=False)  # (flat: the sections merged, utils/common.py:31-39)
    cfg['output_root'] = osp.join(cfg['output_dir'], cfg['output_folder'])
    os.makedirs(osp.join(cfg['output_root'], 'logs'), exist_ok=True)
    cfg['logging'] = get_logger(osp.join(cfg['output_root'], 'logs', 'piecewise_icp_{}.log'.format(time.strftime('%Y%m%d_%H%M%S'))))
    cfg = AttrDict(cfg)
    if args.engine:
        cfg.engine = args.engine
    cfg.device = access_device()
    start = time.time()
    cfg.tile_dir = osp.join(cfg.output_root, 'tiled_data')

    def tiling_config():  # main_piecewise_icp.py:66-78
        c = copy.copy(cfg)
        c.data_dir, c.src_name, c.tgt_name = cfg.input_root, cfg.src_pcd, cfg.tgt_pcd
        return c

    tiles = prepare_tiles(cfg.tile_dir, tiling_config, cfg.logging)
    for_each_tile(cfg, tiles, Piecewise_ICP, cfg.logging)
    cfg.logging.info(f"Displacement estimation is done! Save results to: '{cfg.output_root}'. Total time taken: {time.time() - start:.1f} seconds.")


if __name__ == '__main__':
    main()
