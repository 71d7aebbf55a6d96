"""TEST INFRASTRUCTURE -- a numpy restatement of the reference's tiler, cpp_core/pcd_tiling/pcd_tiling.cpp:709-871 (`tile_point_clouds`)
and :662-707 (`resave_point_cloud`): the checker of fusion4landslide_amd/csrc/tiling.hip (f4l_tile_point_clouds).  Only tests/ import
it.  Rounds 2-5 shipped this code as the product's tiler (host numpy around one device call); since round 6 the product is the
library entry and this file is its oracle.

What it does (same steps, same file names): crop both epochs to the overlap of their bounding boxes (:73-116), thin them with a voxel
grid (pcl::VoxelGrid, :118-227; leaf = median nearest-neighbour spacing of the smaller cloud when voxelGridFilterSize == 0, :37-54),
then halve the bounding box along the longer in-plane side until both halves hold fewer than maxPointsPerTile points (:231-655) and
write every leaf as `non_overlap/{source,target}_tile_<i>.ply` and `overlap/{source,target}_tile_<i>_overlap.ply`, the latter cut from
the parent's overlap cloud with the leaf's box grown by 20 m in the projection plane (the 20 is hard coded in the reference;
`overlapTiles` and `minPointsPerTile` are accepted and unused there too).

PARITY UNPINNED: PCL is not installable in the build container and the reference holds no fixtures for its tiler; the voxel filter
(oracle.voxel_grid_pcl) and the crop follow PCL's documented behaviour.  Centroids are summed in double (PCL: float32
accumulators), colours are averaged like the coordinates.
"""
import os

import numpy as np

from fusion4landslide_amd.utils.ply import read_ply
from . import oracle as O

EPS = 1e-9      # pcd_tiling.cpp:26
_PAD = 20.0     # metres added around a tile for its "overlap" twin (pcd_tiling.cpp:297-301 and every sibling branch)
_f32 = np.float32


class _Cloud:
    """xyz float32 (n, 3) + optional rgb uint8 (n, 3): what pcl::PointXYZRGB keeps of a PLY vertex."""

    def __init__(self, xyz, rgb=None):
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)

    def __len__(self):
        return len(self.xyz)

    def take(self, mask):
        return _Cloud(self.xyz[mask], None if self.rgb is None else self.rgb[mask])


def _load(path):
    xyz, fields = read_ply(path)
    rgb = None
    for names in (("red", "green", "blue"), ("r", "g", "b"), ("diffuse_red", "diffuse_green", "diffuse_blue")):
        if all(k in fields for k in names):
            rgb = np.stack([np.asarray(fields[k]) for k in names], axis=1)
            break
    return _Cloud(xyz, rgb)


def _write(path, cloud):
    """Binary little-endian PLY with float x y z and, when present, uchar red green blue (PLYWriter::write(..., binary
    = true, use_camera = false) of a PointXYZRGB cloud, pcd_tiling.cpp:263-268)."""
    n = len(cloud)
    head = f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\nproperty float x\nproperty float y\nproperty float z\n"
    if cloud.rgb is not None:
        head += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
        rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
        rec["r"], rec["g"], rec["b"] = cloud.rgb[:, 0], cloud.rgb[:, 1], cloud.rgb[:, 2]
    else:
        rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    rec["x"], rec["y"], rec["z"] = cloud.xyz[:, 0], cloud.xyz[:, 1], cloud.xyz[:, 2]
    with open(path, "wb") as f:
        f.write((head + "end_header\n").encode())
        f.write(rec.tobytes())


def _crop(cloud, lo, hi):
    """pcl::CropBox with min / max (pcd_tiling.cpp:104-116): keeps lo <= p <= hi on every axis, in float32."""
    lo, hi = np.asarray(lo, dtype=np.float32), np.asarray(hi, dtype=np.float32)
    keep = np.all((cloud.xyz >= lo) & (cloud.xyz <= hi), axis=1)
    return cloud.take(keep)


def _voxel_grid(cloud, leaf):
    """voxel_grid_filter (pcd_tiling.cpp:118-227).  The reference splits the cloud into octants first when the box would
    hold more than 2^31 leaves, because pcl::VoxelGrid indexes cells with int32; the cell keys here are 64 bit, so the
    filter is applied in one piece (clouds that large differ from the reference at the octant seams, where it filters
    a 2 EPS wide shared slab twice)."""
    if len(cloud) == 0:
        return cloud
    pts, cnt, vop = O.voxel_grid_pcl(cloud.xyz, float(_f32(leaf)))
    out_xyz = np.asarray(pts).astype(np.float32)
    rgb = None
    if cloud.rgb is not None:
        v, c = np.asarray(vop), np.asarray(cnt).astype(np.float64)
        rgb = np.stack([np.bincount(v, weights=cloud.rgb[:, k].astype(np.float64), minlength=len(c)) / c for k in range(3)], axis=1)
        rgb = rgb.astype(np.uint8)  # (the float average is truncated when it is packed back into the rgb field)
    return _Cloud(out_xyz, rgb)


def _median_resolution(cloud):
    """median_point_cloud_resolution (pcd_tiling.cpp:37-54): sqrt of the upper median (index n / 2) of the squared
    distance to the nearest other point."""
    from scipy.spatial import cKDTree
    p = cloud.xyz.astype(np.float64)
    d, _ = cKDTree(p).query(p, k=2)
    d2 = np.sort((d[:, 1] ** 2).astype(np.float32))  # (the squared distances of exactly representable float32 points: exact in double)
    return float(np.sqrt(d2[len(d2) // 2]))


def _split_boxes(lo, hi, direction):
    """One halving step of split_point_clouds_into_tiles (pcd_tiling.cpp:276-655).  Returns two (tile_lo, tile_hi,
    overlap_lo, overlap_hi) tuples, upper half first (the order the reference recurses in).  float32 boxes; the EPS
    terms enter in double and are rounded away again when stored, exactly as `float = float - float / 2 - 1e-9` does."""
    u, v = {0: (1, 2), 1: (0, 2), 2: (0, 1)}[direction]
    side_u, side_v = _f32(hi[u] - lo[u]), _f32(hi[v] - lo[v])
    s, o = (u, v) if side_u > side_v else (v, u)
    half = _f32(_f32(hi[s] - lo[s]) / _f32(2))
    cut = _f32(hi[s] - half)  # float32 like `maxPT.y - side / 2` before the double EPS joins
    parts = []
    for upper in (True, False):
        tlo, thi = lo.copy(), hi.copy()
        olo, ohi = lo.copy(), hi.copy()
        if upper:
            tlo[s] = _f32(np.float64(cut) - EPS)
            olo[s] = _f32(np.float64(cut) - EPS - _PAD)
            ohi[s] = _f32(np.float64(hi[s]) + _PAD)
        else:
            thi[s] = _f32(np.float64(cut) + EPS)
            olo[s] = _f32(np.float64(lo[s]) - _PAD)
            # (one branch of the reference subtracts EPS here instead of adding it: projection along z, split along x,
            #  pcd_tiling.cpp:570 -- below float32 resolution either way)
            sign = -1.0 if (direction == 2 and s == 0) else 1.0
            ohi[s] = _f32(np.float64(cut) + sign * EPS + _PAD)
        olo[o] = _f32(np.float64(lo[o]) - _PAD)
        ohi[o] = _f32(np.float64(hi[o]) + _PAD)
        parts.append((tlo, thi, olo, ohi))
    return parts


def _split(c1, c2, o1, o2, lo, hi, max_pts, counter, direction, save_dir):
    # (:247-248) how many tiles the larger cloud needs at least; 1 = small enough, write it
    if max(len(c1), len(c2)) // max_pts + 1 == 1:
        if min(len(c1), len(c2)) > 1:  # (:253; the 1000-point floor is commented out in the reference)
            i = counter[0]
            _write(os.path.join(save_dir, "non_overlap", f"source_tile_{i}.ply"), c1)
            _write(os.path.join(save_dir, "non_overlap", f"target_tile_{i}.ply"), c2)
            _write(os.path.join(save_dir, "overlap", f"source_tile_{i}_overlap.ply"), o1)
            _write(os.path.join(save_dir, "overlap", f"target_tile_{i}_overlap.ply"), o2)
            counter[0] += 1
        return
    if float(hi[0] - lo[0]) <= 0 and float(hi[1] - lo[1]) <= 0 and float(hi[2] - lo[2]) <= 0:
        raise ValueError("more than maxPointsPerTile coincident points: the box cannot be halved any further "
                         "(the reference recurses until the stack overflows)")
    for tlo, thi, olo, ohi in _split_boxes(lo, hi, direction):
        _split(_crop(c1, tlo, thi), _crop(c2, tlo, thi), _crop(o1, olo, ohi), _crop(o2, olo, ohi), tlo, thi, max_pts,
               counter, direction, save_dir)


def tile_point_clouds(firstPointCloud, secondPointCloud, maxPointsPerTile, minPointsPerTile, voxelGridFlag,
                      voxelGridFilterSize, overlapTiles, projectionDirection, save_dir, verbose):
    say = print if verbose else (lambda *a, **k: None)
    for path in (firstPointCloud, secondPointCloud):
        if not os.path.isfile(path):
            print(f"File {path} does not exist!!!")
            return False  # (:735-738, 750-753)
    if int(maxPointsPerTile) < 1:
        raise ValueError("maxPointsPerTile must be positive")
    c1, c2 = _load(firstPointCloud), _load(secondPointCloud)
    say(f"Point cloud 1 read in! Number of Points: {len(c1)}")
    say(f"Point cloud 2 read in! Number of Points: {len(c2)}")
    if len(c1) == 0 or len(c2) == 0:
        raise ValueError("empty point cloud")
    # overlap of the two bounding boxes and the area of its three faces (:73-102, 763-772)
    lo = np.maximum(c1.xyz.min(axis=0), c2.xyz.min(axis=0)).astype(np.float32)
    hi = np.minimum(c1.xyz.max(axis=0), c2.xyz.max(axis=0)).astype(np.float32)
    ext = (hi - lo).astype(np.float32)
    area = [_f32(ext[1] * ext[2]), _f32(ext[0] * ext[2]), _f32(ext[0] * ext[1])]
    c1, c2 = _crop(c1, lo, hi), _crop(c2, lo, hi)
    for sub in ("", "non_overlap", "overlap"):  # (:800-809; create_directory: no error when it exists)
        os.makedirs(os.path.join(save_dir, sub), exist_ok=True)
    if voxelGridFlag:
        leaf = float(voxelGridFilterSize)
        if leaf == 0.0:  # (:814-821) spacing of the smaller cloud
            leaf = _median_resolution(c1 if len(c1) < len(c2) else c2)
            say(f"Size of the filter: {leaf} m determined based on the median resolution!")
        c1, c2 = _voxel_grid(c1, leaf), _voxel_grid(c2, leaf)
        say(f"{len(c1)} / {len(c2)} points remaining after voxel grid filter.")
    direction = int(projectionDirection)
    if direction == -1:  # (:844-845) project along the axis whose face of the overlap box is largest
        direction = int(np.argmax(area))
    if direction not in (0, 1, 2):
        raise ValueError("projectionDirection must be -1, 0, 1 or 2")
    counter = [0]
    _split(c1, c2, c1, c2, lo, hi, int(maxPointsPerTile), counter, direction, save_dir)
    say(f"Spliting complete. {counter[0]} patches saved per epoch.")
    return True


def resave_point_cloud(firstPointCloud, secondPointCloud, verbose):
    """Re-write both files as binary PLY (:662-707).  The reference loads the second cloud only when `verbose` is set
    (:692-697) and then writes an empty cloud over it otherwise, and falls off the end without a return value; here
    both files are always read and rewritten, and True is returned."""
    for path in (firstPointCloud, secondPointCloud):
        if not os.path.isfile(path):
            print(f"File {path} does not exist!!!")
            return False
    for path in (firstPointCloud, secondPointCloud):
        _write(path, _load(path))
    return True
