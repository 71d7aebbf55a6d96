"""The point-to-plane estimator on the headline cloud (C4), with double and with float32 target normals, beside point-to-point:
the program the counter passes of DESIGN.md section 7 (g) were pointed at (F4L_LIB_PATH selects another build)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
dev = torch.device("cuda")
c = synthetic.CONFIGS["C4_50M_100k"]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
src, soff, tgt, toff, P = d["src"], d["src_off"], d["tgt"], d["tgt_off"], d["P"]
nrm = engine.patch_normals(tgt, toff, 30, max_patch=d["max_tgt"], f64=True)
for name, kw in (("p2plane f64 normals", dict(icp_type="point2plane", tgt_normals=nrm)), ("p2plane f32 normals", dict(icp_type="point2plane", tgt_normals=nrm.float())), ("p2p", dict())):
    f = lambda: engine.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=20, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], **kw)
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(os.environ.get("F4L_LIB_PATH", "default"), name, "%.2f ms" % min(ts), flush=True)
