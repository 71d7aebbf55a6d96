"""Where the dense configuration's time goes (GPU box, repo root): the C3 step under the launch switches, and every size class
of its patches run ALONE (the same patches gathered into a cloud of their own), with points/s per class.

    python3 tools/gpu/c3_diag.py [config] [classes|switches|prof]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from fusion4landslide_amd import engine, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3_10M_20k"
what = sys.argv[2] if len(sys.argv) > 2 else "classes"
dev = torch.device("cuda")
c = synthetic.CONFIGS[cfg]
d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev)
prob = bench.Problem(torch, engine, synthetic, d, dev)
n_src = int(d["src"].shape[0])


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def subset(ids):
    """The patches `ids` (and their Kabsch correspondences) as a problem of their own."""
    def gather(pts, off):
        cnt = (off[1:] - off[:-1])[ids]
        noff = torch.zeros(ids.shape[0] + 1, dtype=torch.int64, device=dev)
        noff[1:] = torch.cumsum(cnt, 0)
        start = off[:-1][ids]
        idx = torch.repeat_interleave(start - noff[:-1], cnt) + torch.arange(int(noff[-1]), device=dev)
        return pts[idx].contiguous(), noff, int(cnt.max()) if ids.numel() else 0
    s, so, ms = gather(d["src"], d["src_off"])
    t, to, mt = gather(d["tgt"], d["tgt_off"])
    cs, coff, _ = gather(prob.cs, prob.coff)
    ct, _, _ = gather(prob.ct, prob.coff)
    return dict(src=s, src_off=so, tgt=t, tgt_off=to, cs=cs, ct=ct, coff=coff, max_src=ms, max_tgt=mt)


def run(q):
    return engine.patch_loop(q["src"], q["src_off"], q["tgt"], q["tgt_off"], q["cs"], q["ct"], q["coff"], None, 0.0, 1e-6,
                             max_corr_dist=bench.MAX_CORR, max_iter=bench.MAX_ITER, fixed_iters=True, max_src_patch=q["max_src"],
                             max_tgt_patch=q["max_tgt"], search="f64")


if what == "switches":
    for env in ({}, {"F4L_ICP_SERIAL_CLASSES": "1"}, {"F4L_ICP_NOCLASSES": "1"}, {"F4L_ICP_THROUGHPUT": "1"}, {"F4L_ICP_THROUGHPUT": "0"},
                {"F4L_ICP_CLASS_STEP": "2"}, {"F4L_ICP_SUBDIV": "4"}, {"F4L_ICP_SUBDIV": "16"}, {"F4L_ICP_DENS": "4"}, {"F4L_ICP_DENS": "1"},
                {"F4L_ICP_DEBUG": "4"}, {"F4L_ICP_DEBUG": "16"}, {"F4L_ICP_NOPP": "1"}, {"F4L_ICP_MU": "0.03"}, {"F4L_ICP_MU": "0.5"}):
        for k, v in env.items():
            os.environ[k] = v
        ms = timed(prob.step)
        for k in env:
            del os.environ[k]
        print(f"{str(env):45s} {ms:8.2f} ms  {n_src / ms / 1e3:8.1f} Mpts/s", flush=True)
elif what == "classes":
    ns = d["src_off"][1:] - d["src_off"][:-1]
    nt = d["tgt_off"][1:] - d["tgt_off"][:-1]
    big = torch.maximum(ns, nt)
    out = prob.step()
    fit = out["fitness"]
    print("whole step", round(timed(prob.step), 2), "ms")
    lo = -1
    tot = 0.0
    for hi in (0, 64, 128, 256, 384, 512, 640, 768, 1024, 1536, 2048, 3072, 1 << 30):
        ids = torch.nonzero((big > lo) & (big <= hi), as_tuple=True)[0]
        lo = hi
        if ids.numel() == 0:
            continue
        q = subset(ids)
        ms = timed(lambda: run(q), reps=3, warm=1)
        tot += ms
        npts = int(q["src"].shape[0])
        print(f"class <= {hi:10d}: {ids.numel():6d} patches, mean ns {npts / ids.numel():7.1f} nt {int(q['tgt'].shape[0]) / ids.numel():7.1f} "
              f"fitness {float(fit[ids].mean()):.3f}: {ms:7.2f} ms alone, {ms / ids.numel() * 1e3:7.2f} us/patch, {npts / ms / 1e3:8.1f} Mpts/s", flush=True)
    print("sum of the classes alone", round(tot, 2), "ms")
    # by fitness: patches whose block moved out of reach (no correspondences) against registered ones
    for lo_f, hi_f in ((-1, 0.01), (0.01, 0.5), (0.5, 0.9), (0.9, 2)):
        ids = torch.nonzero((fit > lo_f) & (fit <= hi_f) & (big <= 768), as_tuple=True)[0]
        if ids.numel() == 0:
            continue
        q = subset(ids)
        ms = timed(lambda: run(q), reps=3, warm=1)
        print(f"bulk, fitness ({lo_f}, {hi_f}]: {ids.numel():6d} patches: {ms:7.2f} ms alone, {ms / ids.numel() * 1e3:7.2f} us/patch", flush=True)
else:  # profiling build: phase cycles of the bulk patches (F4L_LIB_PATH must point at the -DF4L_ICP_PROF library)
    os.environ["F4L_ICP_PROF"] = "1"
    os.environ["F4L_ICP_DEBUG"] = "64"
    ns = d["src_off"][1:] - d["src_off"][:-1]
    nt = d["tgt_off"][1:] - d["tgt_off"][:-1]
    big = torch.maximum(ns, nt)
    for lo, hi in ((0, 512), (512, 768), (768, 1536)):
        ids = torch.nonzero((big > lo) & (big <= hi), as_tuple=True)[0]
        q = subset(ids)
        print("class", lo, hi, ids.numel(), flush=True)
        run(q)
        torch.cuda.synchronize()
