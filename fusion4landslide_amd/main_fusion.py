"""Counterpart of the reference's main_fusion.py for the 3D hot path: `--config` (nested yaml, main_fusion.py:63) -> cfg -> tiling
once -> per tile `Coarse2Fine(cfg).implement_c2f_matching()` (:134-148) -> `results/c2f_*_tile_<id>.txt`.

    python -m fusion4landslide_amd.main_fusion --config configs/landslide/fusion_3d_brienz.yaml [--partition identical]

The supervoxel partition of this entry is the REFERENCE's own (`identical`: f4l_supervoxel -- kNN, normals and the sequential
fusion / FIFO exchange of supervoxel_segmentation.h:117-237 on the device, label for label: 30 ms per 1 M-point tile), so a
drop-in run writes the partition and `c2f_*_tile` files the reference writes.  `--partition parallel` -- or F4L_SV_MODE=parallel --
opts into the parallel variant (f4l_supervoxel_parallel: the reference's K, criteria and partition quality in 4 ms, not its labels).
The mode is validated before anything runs and logged at start-up.

The config keys are the reference's (path_name / data / method / parameter_setting / misc).  What the reference computes with
its learned models -- point matches, patch matches, the 2D matches lifted to 3D -- enters through hooks on the cfg
(src/coarse_to_fine_matching.py of this package); without them the 3D stand-ins run (nearest neighbours), and
`fine_matching_fusion` / `_only_2d` need `cfg.point_matches_from_2d`.  `method.partition_type` must be `supervoxel`.

Steps of the reference's main() that are NOT done here, on purpose: loading the DIP weights (main_fusion.py:35-45: the learned
descriptor is out of scope), the `project_dir` setting and the dump of the whole config into the log (:66-104), and the
end-of-run clean-up that deletes the `processed` / `raw` folders of the data directory (:150-153) -- this entry never deletes
anything it did not write.
"""
import argparse
import copy
import os
import os.path as osp
import time

from . import engine
from .src.coarse_to_fine_matching import Coarse2Fine
from .utils import async_io
from .utils.common import AttrDict, access_device, get_logger, load_yaml, setup_seed
from .utils.tiles import for_each_tile, prepare_tiles


def build_config(path, log_prefix='coarse2fine_matching'):
    cfg = load_yaml(path, keep_sub_directory=True)
    cfg['path_name']['output_root'] = osp.join(cfg['path_name']['output_dir'], cfg['path_name']['output_folder'])
    log_dir = osp.join(cfg['path_name']['output_root'], 'logs')
    os.makedirs(log_dir, exist_ok=True)
    log_path = osp.join(log_dir, '{}_{}.log'.format(log_prefix, time.strftime('%Y%m%d_%H%M%S')))
    cfg['logging'] = get_logger(log_path)
    cfg = AttrDict(cfg)
    cfg.verbose = cfg.misc.verbose
    cfg.save_interim = cfg.misc.save_interim
    cfg.device = access_device()
    return cfg, log_path


def run(cfg, first_tile=0, tiles_per_launch=1):
    """main_fusion.py:106-148 on a prepared cfg (also the entry for callers that attach the matching hooks).  `tiles_per_launch`
    tiles share one per-patch launch (engine.patch_loop_tiles; 1 = the reference's tile-by-tile order of work; results agree to
    rounding, 1e-9 m in a transform, whatever the batch)."""
    import torch
    tile_dir = cfg.path_name.tile_dir = osp.join(cfg.path_name.output_root, 'tiled_data')

    def tiling_config():  # main_fusion.py:113-123
        c = copy.copy(cfg)
        c.data_dir, c.src_name, c.tgt_name = cfg.path_name.input_root, cfg.data.src_pcd, cfg.data.tgt_pcd
        c.tiling_type, c.max_pts_per_tile, c.min_pts_per_tile = cfg.method.tiling_type, cfg.method.max_pts_per_tile, cfg.method.min_pts_per_tile
        c.voxel_size, c.tile_dir = cfg.method.voxel_size_init, tile_dir
        return c

    # host files next to device work (utils/async_io.py): the partition and result files of a tile are written by writer threads
    # while the next tile computes, and the next tile's PLY files are read ahead; the same bytes land on disk, and every writer is
    # through (or has raised) before this function returns.  F4L_ASYNC_IO=0: the reference's serial order of work.
    cfg.defer_files = async_io.enabled()
    try:
        with torch.no_grad():
            tiles = _run_tiles(cfg, tile_dir, tiling_config, first_tile, tiles_per_launch)
    finally:
        async_io.drain()
    engine.release_scratch()  # (the run's peak workspace does not outlive it)
    return tiles


def _run_tiles(cfg, tile_dir, tiling_config, first_tile, tiles_per_launch):
    tiles = prepare_tiles(tile_dir, tiling_config, cfg.logging)

    def launch(states):
        kw = states[0].fine_state["loop_kw"]
        assert all(s.fine_state["loop_kw"] == kw for s in states)  # (one config: one set of loop parameters)
        return engine.patch_loop_tiles([s.fine_state["loop_args"] for s in states], **kw)

    for_each_tile(cfg, tiles, lambda c: Coarse2Fine(c).implement_c2f_matching(), cfg.logging, first=first_tile, batch=int(tiles_per_launch),
                  stages=(lambda c: Coarse2Fine(c).prepare_c2f(), launch, lambda s, out: s.finish_c2f(out)))
    return tiles


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='./configs/landslide/fusion_brienz.yaml',  # main_fusion.py:57-63
                        help="Path to a fusion config of the reference's layout; `method.partition_type` must be `supervoxel`.")
    parser.add_argument('--partition', type=str, default=None, choices=['identical', 'parallel'],
                        help="supervoxel segmentation: the reference's labels (default) or the all-device segmentation")
    parser.add_argument('--first-tile', type=int, default=0)
    parser.add_argument('--tiles-per-launch', type=int, default=1,
                        help="tiles whose per-patch loops run as ONE launch: 1 (default) = tile by tile like the reference; more is faster "
                             "(1.5 x the loop's rate at 8) and agrees to rounding -- a '%%.6f' row may differ in its last digit")
    args = parser.parse_args(argv)
    mode = args.partition or os.environ.get("F4L_SV_MODE", "identical")
    if mode not in ("identical", "parallel"):  # (before tiling starts, not inside the first computeSupervoxel)
        parser.error(f"F4L_SV_MODE must be 'identical' or 'parallel', not {mode!r}")
    setup_seed(0)
    cfg, log_path = build_config(args.config)
    from .cpp_core.supervoxel_segmentation.build import supervoxel
    mode_before = supervoxel.SEGMENTATION
    supervoxel.SEGMENTATION = mode
    cfg.logging.info(f"supervoxel partition mode: {mode!r} " + ("(the reference's labels)" if mode == "identical" else
                     "(device segmentation: the reference's K and criteria, NOT its labels; partition files differ from the reference's)"))
    start = time.time()
    try:
        run(cfg, args.first_tile, args.tiles_per_launch)
    finally:
        supervoxel.SEGMENTATION = mode_before  # (the module's own default is its callers' business, not this entry's)
    if cfg.verbose:
        cfg.logging.info(f"Displacement estimation is done! Save log information to: '{log_path}'.")
        cfg.logging.info(f"Save results to: '{cfg.path_name.output_root}'. Total time taken: {time.time() - start:.1f} seconds.")


if __name__ == '__main__':
    main()
