"""The batched loop body of fine_matching_with_different_types (fusion4landslide_amd.src.fine_matching.fine_matching_3d: mutual
gather, quality check, Kabsch, ICP with the reference criteria, rows, tgt2src rows, nearest-neighbour refinement) on the C2
tile and on 16 of them, wall time per call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
from fusion4landslide_amd.src.fine_matching import fine_matching_3d
dev = torch.device("cuda")
for name in ("C2_1M_2k", "C2x16_16M_32k"):
    c = synthetic.CONFIGS[name]
    d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
    src, soff, tgt, toff, P = d["src"], d["src_off"], d["tgt"], d["tgt_off"], d["P"]
    n = src.shape[0]
    eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
    thr = torch.full((P,), 0.2, dtype=torch.float64, device=dev)
    nn, _ = engine.nn_refine(src, soff, tgt, toff, eye, thr, max_tgt_patch=d["max_tgt"], return_rows=False)
    pid = torch.repeat_interleave(torch.arange(P, device=dev), soff[1:] - soff[:-1])
    corr = torch.where(nn >= 0, toff[pid] + nn.to(torch.int64), torch.full_like(pid, -1))
    sid, tid = torch.arange(n, device=dev), torch.arange(tgt.shape[0], device=dev)
    for kw in (dict(remove_low_quality_patch_matches=False), dict(remove_low_quality_patch_matches=True), dict(remove_low_quality_patch_matches=True, rigidity_precision="f32"), dict(remove_low_quality_patch_matches=True, assign_type="assign_then_nn", output_tgt2src=True, median_max_resolution=0.03)):
        f = lambda: fine_matching_3d(src, tgt, sid, soff, tid, toff, corr, thres_dist_diff=0.03, **kw)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{name} fine_matching_3d {kw}: {min(ts):.2f} ms wall; registered {int((r['iters'] >= 0).sum())} of {P}", flush=True)
    del d
    torch.cuda.empty_cache()
