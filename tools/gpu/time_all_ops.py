"""Every entry point of the library around the hot loop, timed alone on the C2 tile (1 M points per epoch, 2025 patches of
~500 points; C4 for the ones that scale with the patch count): a table for DESIGN.md, and a way to spot a kernel that is
off by an order of magnitude.  Usage: time_all_ops.py [config]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2_1M_2k"
dev = torch.device("cuda")
c = synthetic.CONFIGS[cfg]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
src, soff, tgt, toff, P = d["src"], d["src_off"], d["tgt"], d["tgt_off"], d["P"]
n = src.shape[0]


def timed(name, fn, reps=5, unit_pts=n):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); out = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = min(ts)
    print(f"{name:58s} {ms:9.3f} ms  {unit_pts / ms / 1e3:9.1f} M pts/s", flush=True)
    return out


eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
thr = torch.full((P,), 0.2, dtype=torch.float64, device=dev)
nn, _ = engine.nn_refine(src, soff, tgt, toff, eye, thr, max_tgt_patch=d["max_tgt"], return_rows=False)
cs, ct, coff = synthetic.correspondences_from_nn_device(src, soff, tgt, toff, nn)
m = cs.shape[0]
print(f"{cfg}: {n} points per epoch, {P} patches, {m} point matches")
R, t = timed("kabsch_batched (float32 in, 3x3 + t out)", lambda: engine.kabsch_batched(cs, ct, coff), unit_pts=m)
T0 = timed("kabsch_transforms (4x4 out)", lambda: engine.kabsch_transforms(cs, ct, coff), unit_pts=m)
timed("kabsch_residuals", lambda: engine.kabsch_residuals(cs, ct, coff, R, t), unit_pts=m)
timed("kabsch2_batched (src/functions.py Kabsch #2)", lambda: engine.kabsch2_batched(cs, ct, coff), unit_pts=m)
timed("rigidity_check (all pairs of every match set)", lambda: engine.rigidity_check(cs, ct, coff, 0.03), unit_pts=m, reps=3)
timed("rigidity_check, float32 pair arithmetic", lambda: engine.rigidity_check(cs, ct, coff, 0.03, precision="f32"), unit_pts=m, reps=3)
timed("apply_transform (rows [s, T s])", lambda: engine.apply_transform(src, soff, T0))
timed("nn_refine (refine_dvfs_with_threshold)", lambda: engine.nn_refine(src, soff, tgt, toff, T0, thr, max_tgt_patch=d["max_tgt"]))
timed("patch_normals (estimate_normals per patch, k = 30)", lambda: engine.patch_normals(tgt, toff, 30, max_patch=d["max_tgt"]), reps=3)
timed("piecewise_icp point2point, 20 fixed iterations, from T0", lambda: engine.piecewise_icp(src, soff, tgt, toff, init_T=T0, max_iter=20, fixed_iters=True,
                                                                                               max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"]))
timed("patch_loop (Kabsch -> ICP(20) -> rows, one launch)", lambda: engine.patch_loop(src, soff, tgt, toff, cs, ct, coff, None, 0.0, 1e-6, max_iter=20, fixed_iters=True,
                                                                                   max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"]))
if n <= 20_000_000:
    timed("knn k = 30", lambda: engine.knn(src, 30))
    timed("knn_normals k = 30", lambda: engine.knn_normals(src, 30))
    idx = engine.knn(src, 30)
    timed("normals (from given lists)", lambda: engine.normals(src, idx))
    timed("nn_query k = 1 (epoch 2 against epoch 1)", lambda: engine.nn_query(src, tgt, 1))
    timed("median_resolution (both epochs)", lambda: engine.median_resolution(src, tgt))
    timed("voxel_downsample (0.1 m)", lambda: engine.voxel_downsample(src, 0.1))
    lab, K = engine.supervoxel_parallel(src, 30, 0.526)
    timed("supervoxel_parallel (res 0.526 m)", lambda: engine.supervoxel_parallel(src, 30, 0.526), reps=3)
    timed("labels_to_csr", lambda: engine.labels_to_csr(lab, K))
    order, off = engine.labels_to_csr(lab, K)
    timed("gather_points", lambda: engine.gather_points(src, order))
    ids = order.to(torch.int64)
    corr_tgt = torch.arange(n, device=dev, dtype=torch.int64)
    timed("mutual_correspondences (isin gather)", lambda: engine.mutual_correspondences(ids, off, ids, off, corr_tgt))
