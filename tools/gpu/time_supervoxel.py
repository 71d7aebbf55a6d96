"""Stage times of the supervoxel partition of a 1 M-point tile: label-identical mode (kNN + normals on the device, sequential
segmentation on one host core) against the parallel mode (everything on the device)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cells = int(round(45 * (n / 1e6) ** 0.5))
d = synthetic.make_patches_device(n, cells, 1.386, torch.device("cuda"), seed=0)
xyz, k, res = d["src"], 30, 1.386
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), r
ms_knn, knn = ev(lambda: engine.knn(xyz, k))
ms_nrm, nrm = ev(lambda: engine.normals(xyz, knn))
ms_seg, (labels, info, reps) = ev(lambda: engine.supervoxel_segment_device(xyz, nrm, knn, res, return_reps=True))
info = info.cpu().numpy()
print(f"n={n}: knn {ms_knn:.2f} ms, normals {ms_nrm:.2f} ms, device segmentation {ms_seg:.2f} ms (K={info[0]}, target {info[1]}, status {info[2]}, sweeps {info[3]})")
t0 = time.perf_counter(); lab_p, Kp = engine.supervoxel_parallel(xyz, k, res); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"f4l_supervoxel_parallel end to end: {1e3*(t1-t0):.1f} ms, K={Kp}")
t0 = time.perf_counter(); lab_h, Kh = engine.supervoxel(xyz, k, res); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"f4l_supervoxel (the reference's labels, on the device) end to end: {1e3*(t1-t0):.1f} ms, K={Kh}")
cnt_p, cnt_h = torch.bincount(lab_p.long()).float(), torch.bincount(lab_h.long()).float()
print(f"sizes parallel: min {int(cnt_p.min())} max {int(cnt_p.max())} cv {float(cnt_p.std()/cnt_p.mean()):.3f}; sequential: min {int(cnt_h.min())} max {int(cnt_h.max())} cv {float(cnt_h.std()/cnt_h.mean()):.3f}")
