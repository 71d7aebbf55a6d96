"""f4l_supervoxel on ONE cloud in several point orders: the bench's (patch by patch, random inside a patch), a voxel-grid filter's
(sorted by voxel index: x fastest, then y, then z -- what pcl::VoxelGrid and the tiler write), Morton, and a random shuffle.
Passes (F4L_SV_EXACT_DEBUG=1 on stderr) and time per order.  Usage: svx_orders.py [n_points]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))
lo = xyz.min(0).values
cell = ((xyz - lo) / 0.05).floor().to(torch.int64)  # (5 cm voxels: one or two points each)
nx, ny = int(cell[:, 0].max()) + 1, int(cell[:, 1].max()) + 1
def part1by1(v):
    v = v & 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
    return v
orders = {"patch by patch (the bench's)": None,
          "voxel index, x fastest (a voxel-grid filter's output)": torch.argsort((cell[:, 2] * ny + cell[:, 1]) * nx + cell[:, 0], stable=True),
          "Morton (x, y)": torch.argsort(part1by1(cell[:, 0]) | (part1by1(cell[:, 1]) << 1), stable=True),
          "random shuffle": torch.randperm(n, device=xyz.device, generator=torch.Generator(device=xyz.device).manual_seed(1))}
for name, o in orders.items():
    p = xyz if o is None else xyz[o].contiguous()
    print(f"== {name}", file=sys.stderr, flush=True)
    engine.supervoxel(p, 30, res); ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); labels, K = engine.supervoxel(p, 30, res); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name}: {n} points, {min(ts):.2f} ms, K = {K}", flush=True)
