"""The part the reference's entry scripts share (main_piecewise_icp.py:60-94, main_fusion.py:106-148): fill
`<output_root>/tiled_data` through `point_cloud_tiling` when it is empty, then visit the overlap tiles in numeric order, setting
`tile_id`, `src_tile_overlap_path` and `tgt_tile_overlap_path` on the config before each tile is processed."""
import glob
import os
import os.path as osp
import re

from . import async_io
from .common import dir_exist
from .ply import read_xyz32


def prepare_tiles(tile_dir, tiling_config, log):
    """Runs the tiler into an empty `tile_dir` (src/functions.py:147-177; tiles already there are used as they are) and returns the
    source overlap tiles sorted by tile number."""
    from ..src.functions import point_cloud_tiling
    dir_exist(tile_dir)
    if not any(os.listdir(tile_dir)):
        point_cloud_tiling(tiling_config())
    else:
        log.info('Skip point cloud tiling. Tiles will be loaded from %s.', tile_dir)
    tiles = sorted(glob.glob(osp.join(tile_dir, 'overlap', 'source_tile_*')), key=lambda x: int(re.search(r'\d+', osp.basename(x)).group()))
    log.info(f'Num. of tile(s) from source/target point cloud: {len(tiles)}')
    return tiles


def for_each_tile(cfg, tiles, process, log, first=0, batch=1, stages=None):
    """`process(cfg)` per tile, `first` being the reference's hand-edited `continue_tile`.

    With `stages = (prepare, launch, finish)` and `batch` > 1 the tiles are worked `batch` at a time around ONE per-patch launch:
    `prepare(cfg)` per tile (everything before the loop; returns the tile's state), `launch(states)` once for the batch (returns one
    loop result per state), `finish(state, result)` per tile in tile order.  A <= 1 M-point tile's patch matches fill an MI355X for
    two rounds of workgroups; merged launches run at 1.5 x the rate (bench.py extras.C2x8_tiles).  Every patch's result is what its
    own launch gives to rounding (a larger batch may run in another launch shape, whose sums run in another order: 1e-9 m in a
    transform, tests/test_gpu_parity.py), so a '%.6f' row of the files can differ in its last digit."""
    todo = tiles[first:]
    ahead = bool(getattr(cfg, "defer_files", False))  # (the fusion entry: the next tile's PLY files are read while this one computes)

    def visit(tile_i, src_path):
        log.info(f'Current tile {tile_i + first} of total {len(tiles)} tiles')
        tgt_path = src_path.replace('source_tile_', 'target_tile_')
        assert osp.exists(tgt_path), tgt_path
        cfg.tile_id = re.findall(r'\d+', osp.basename(src_path))[0]
        cfg.src_tile_overlap_path, cfg.tgt_tile_overlap_path = src_path, tgt_path
        if ahead and tile_i + 1 < len(todo):
            nxt = todo[tile_i + 1]
            for path in (nxt, nxt.replace('source_tile_', 'target_tile_')):
                if osp.exists(path):
                    async_io.prefetch(path, read_xyz32)

    if stages is None or batch <= 1:
        try:
            for tile_i, src_path in enumerate(todo):
                visit(tile_i, src_path)
                process(cfg)
        finally:
            async_io.forget_prefetched()
        return
    prepare, launch, finish = stages
    pending = []

    def flush():
        if pending:
            for state, result in zip(pending, launch(pending)):
                finish(state, result)
            pending.clear()

    try:
        for tile_i, src_path in enumerate(todo):
            visit(tile_i, src_path)
            pending.append(prepare(cfg))
            if len(pending) >= batch:
                flush()
    finally:
        async_io.forget_prefetched()
        # (also when prepare() raises at tile i: the tiles prepared before it are finished and written, as the reference's
        #  tile-by-tile loop would have written them before failing at i -- ADVICE r5)
        flush()
