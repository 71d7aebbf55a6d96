"""Experiment: how much of the device segmentation's time is the ORDER of the points in memory?  The same 1 M-point tile in the
order the generator gives (patch-contiguous, arbitrary inside a 1.386 m patch) and re-ordered along a fine raster of cells
(~16 points per cell, cells row by row): kNN + normals recomputed on each, then f4l_supervoxel_segment_device timed.
Labels differ between the two (the variant's coins and ties depend on point indices); only the times are compared."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cells = int(round(45 * (n / 1e6) ** 0.5))
d = synthetic.make_patches_device(n, cells, 1.386, torch.device("cuda"), seed=0)
xyz0 = d["src"]


def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), r


def reorder(xyz, cell):
    cx = torch.floor(xyz[:, 0].double() / cell).long()
    cy = torch.floor(xyz[:, 1].double() / cell).long()
    key = cy * (int(cx.max()) + 1) + cx
    return xyz[torch.argsort(key, stable=True)].contiguous()


for res in (1.386, 0.5738):
    for name, xyz in (("generator order", xyz0), ("raster 0.25 m", reorder(xyz0, 0.25)), ("raster 0.5 m", reorder(xyz0, 0.5)),
                      ("random order", xyz0[torch.randperm(n, device="cuda")].contiguous())):
        k = 30
        knn = engine.knn(xyz, k)
        nrm = engine.normals(xyz, knn)
        ms, (labels, info) = ev(lambda: engine.supervoxel_segment_device(xyz, nrm, knn, res))
        info = info.cpu().numpy()
        print(f"res {res}: {name:16s} segmentation {ms:6.2f} ms  K={info[0]} status={info[2]} sweeps={info[3]}", flush=True)
