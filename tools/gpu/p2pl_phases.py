"""Point-to-plane against point-to-point on the headline cloud (C4): time, and with a profiling build (tools/build_variant.sh icp_prof
PROF=1; F4L_LIB_PATH) the per-phase shader-clock shares the library prints per launch (`[icp prof]` on stderr).
F4L_ICP_DEBUG=2048: the plane sums where the points are found (before round 6) instead of in a loop of their own.
Usage: p2pl_phases.py [config]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "C4_50M_100k"
c = synthetic.CONFIGS[name]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
src, soff, tgt, toff = d["src"], d["src_off"], d["tgt"], d["tgt_off"]
nrm = engine.patch_normals(tgt, toff, 30, max_patch=d["max_tgt"], f64=True)
prof = "prof" in os.environ.get("F4L_LIB_PATH", "")
for label, kw, dbg in (("point2plane, summed in its own loop", dict(icp_type="point2plane", tgt_normals=nrm), None),
                       ("point2plane, summed where found (F4L_ICP_DEBUG=2048)", dict(icp_type="point2plane", tgt_normals=nrm), "2048"),
                       ("point2point", dict(), None)):
    if dbg: os.environ["F4L_ICP_DEBUG"] = dbg
    else: os.environ.pop("F4L_ICP_DEBUG", None)
    f = lambda: engine.piecewise_icp(src, soff, tgt, toff, max_corr_dist=0.1, max_iter=20, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], **kw)
    f(); torch.cuda.synchronize(); ts = []
    if prof:
        os.environ["F4L_ICP_PROF"] = "1"
        print(f"== {label}", file=sys.stderr, flush=True)
        f(); torch.cuda.synchronize()
        del os.environ["F4L_ICP_PROF"]
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name} {label}: {min(ts):.2f} ms", flush=True)
os.environ.pop("F4L_ICP_DEBUG", None)
