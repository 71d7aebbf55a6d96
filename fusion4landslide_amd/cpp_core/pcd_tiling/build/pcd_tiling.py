"""Drop-in for the SWIG module `cpp_core.pcd_tiling.build.pcd_tiling` of the reference (cpp_core/pcd_tiling/pcd_tiling.i,
body cpp_core/pcd_tiling/pcd_tiling.cpp), as imported at src/functions.py:5 and main_f2s3.py:4:

    tile_point_clouds(firstPointCloud, secondPointCloud, maxPointsPerTile, minPointsPerTile, voxelGridFlag,
                      voxelGridFilterSize, overlapTiles, projectionDirection, save_dir, verbose) -> bool   :709-871
    resave_point_cloud(firstPointCloud, secondPointCloud, verbose) -> bool                                 :662-707

A shim over the library entries f4l_tile_point_clouds / f4l_resave_point_cloud (csrc/tiling.hip, include/f4l.h): the same ten
arguments in the same order, the same file names.  The tiler itself -- crop to the overlap of the bounding boxes (:73-116), voxel grid
(pcl::VoxelGrid, :118-227; leaf = median nearest-neighbour spacing of the smaller cloud when voxelGridFilterSize == 0, :37-54),
halving of the box along the longer in-plane side until both halves hold fewer than maxPointsPerTile points (:231-655), the
`non_overlap/{source,target}_tile_<i>.ply` and `overlap/{source,target}_tile_<i>_overlap.ply` files (the leaf's box grown by 20 m in the
projection plane: hard coded in the reference; `overlapTiles` and `minPointsPerTile` are accepted and unused there too) -- runs
behind the C ABI with both clouds on the device (rounds 2-5: host numpy around one device call; that code is now the checker,
oracle/pcd_tiling_ref.py).  No CPU fallback.

PCL is not installable in the build container: the voxel filter and the crop follow PCL's documented behaviour [3P-knowledge,
parity unpinned].  Differences from the reference, on purpose: an unreadable or empty cloud raises instead of crashing later, more
than maxPointsPerTile coincident points raise instead of overflowing the stack, and `resave_point_cloud` always rewrites both files.
"""
import ctypes as C

from ...._lib import F4LError, check, lib, require_gpu, stream_ptr


def tile_point_clouds(firstPointCloud, secondPointCloud, maxPointsPerTile, minPointsPerTile, voxelGridFlag,
                      voxelGridFilterSize, overlapTiles, projectionDirection, save_dir, verbose):
    if int(maxPointsPerTile) < 1:
        raise ValueError("maxPointsPerTile must be positive")
    if int(projectionDirection) not in (-1, 0, 1, 2):
        raise ValueError("projectionDirection must be -1, 0, 1 or 2")
    require_gpu()
    n = C.c_int32(-1)
    rc = lib().f4l_tile_point_clouds(str(firstPointCloud).encode(), str(secondPointCloud).encode(), int(maxPointsPerTile), int(minPointsPerTile),
                                     1 if voxelGridFlag else 0, float(voxelGridFilterSize), float(overlapTiles), int(projectionDirection),
                                     str(save_dir).encode(), 1 if verbose else 0, C.byref(n), stream_ptr())
    if rc == -4:  # F4L_EUNSUPPORTED
        raise ValueError("tile_point_clouds: more than maxPointsPerTile coincident points -- the box cannot be halved any further (the "
                         "reference recurses until the stack overflows) -- or a PLY with list properties on its vertex element")
    if rc == -1:  # F4L_EINVAL
        raise ValueError("tile_point_clouds: an input is not a readable PLY file with x, y, z vertex properties, or holds no points")
    check(rc, "f4l_tile_point_clouds")
    if n.value == -2:
        for path in (firstPointCloud, secondPointCloud):
            import os
            if not os.path.isfile(path):
                print(f"File {path} does not exist!!!")
        return False  # (:735-738, 750-753)
    return True


def resave_point_cloud(firstPointCloud, secondPointCloud, verbose):
    """Re-write both files as binary PLY (:662-707).  The reference loads the second cloud only when `verbose` is set (:692-697) and
    then writes an empty cloud over it otherwise, and falls off the end without a return value; here both files are always read and
    rewritten, and True is returned."""
    ok = C.c_int32(0)
    rc = lib().f4l_resave_point_cloud(str(firstPointCloud).encode(), str(secondPointCloud).encode(), 1 if verbose else 0, C.byref(ok))
    if rc == -1:
        raise ValueError("resave_point_cloud: an input is not a readable PLY file with x, y, z vertex properties")
    check(rc, "f4l_resave_point_cloud")
    if not ok.value:
        import os
        for path in (firstPointCloud, secondPointCloud):
            if not os.path.isfile(path):
                print(f"File {path} does not exist!!!")
        return False
    return True


__all__ = ["tile_point_clouds", "resave_point_cloud", "F4LError"]
