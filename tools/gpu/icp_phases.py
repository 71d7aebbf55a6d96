"""Per-phase shader-clock shares of icp_kernel (profiling build: tools/build_variant.sh icp_prof PROF=1, F4L_LIB_PATH pointing
at it), size class by size class, every class gathered into a cloud of its own so that its launch gets the class's own LDS plan.

    F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prof.so python3 tools/gpu/icp_phases.py C4_50M_100k     # bulk / border classes of a config
    F4L_LIB_PATH=...                               python3 tools/gpu/icp_phases.py tile [n]        # the supervoxel patches of a tile
The library prints one `[icp prof]` line per launch on stderr (mean cycles per workgroup: build, phase1 = certify sweep, search,
reduce = row sums + barrier, solve, barrier = wait for the solve)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from fusion4landslide_amd import engine, pipeline, synthetic  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "C4_50M_100k"
dev = torch.device("cuda")
os.environ["F4L_ICP_PROF"] = "1"
if os.environ.get("ICP_PHASES_COUNTERS"):  # per-query search counters as well: same-address atomics, the cycle shares are then distorted
    os.environ["F4L_ICP_DEBUG"] = "64"


def gather(pts, off, ids):
    cnt = (off[1:] - off[:-1])[ids]
    noff = torch.zeros(ids.shape[0] + 1, dtype=torch.int64, device=dev)
    noff[1:] = torch.cumsum(cnt, 0)
    idx = torch.repeat_interleave(off[:-1][ids] - noff[:-1], cnt) + torch.arange(int(noff[-1]), device=dev)
    return pts[idx].contiguous(), noff, int(cnt.max()) if ids.numel() else 0


def run_class(src, so, tgt, to, cs, ct, coff, ids, label, env=None):
    s, soff, ms = gather(src, so, ids)
    t, toff, mt = gather(tgt, to, ids)
    a, aoff, _ = gather(cs, coff, ids)
    b, _, _ = gather(ct, coff, ids)
    for k, v in (env or {}).items():
        os.environ[k] = v
    print(f"class {label}: {ids.numel()} patches, {int(soff[-1])} source points, largest {ms} / {mt}, env {env or {}}", file=sys.stderr, flush=True)
    engine.patch_loop(s, soff, t, toff, a, b, aoff, None, 0.0, 1e-6, max_corr_dist=bench.MAX_CORR, max_iter=bench.MAX_ITER,
                      fixed_iters=True, max_src_patch=ms, max_tgt_patch=mt, search="f64")
    torch.cuda.synchronize()
    for k in (env or {}):
        del os.environ[k]


if what == "tile":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    c = synthetic.make_patches_device(n, 45, 1.386, dev, seed=0)
    r = pipeline.full_path(c["src"], c["tgt"], max_iter=20, fixed_iters=True, keep_inputs=True)
    so, to = r["src_off"], r["tgt_off"]
    ns = so[1:] - so[:-1]
    nt = to[1:] - to[:-1]
    big = torch.maximum(ns, nt)
    print("patches", so.shape[0] - 1, "mean", float(ns.float().mean()), "max", int(ns.max()), file=sys.stderr)
    for w, lo, hi in ((1, 0, 64), (2, 64, 128), (4, 128, 1 << 30)):
        ids = torch.nonzero((big > lo) & (big <= hi), as_tuple=True)[0]
        if ids.numel():
            run_class(r["patch_src"], so, r["patch_tgt"], to, r["corr_src"], r["corr_ref"], r["corr_off"], ids, f"({lo}, {hi}]",
                      {"F4L_ICP_WAVES": str(w)})
else:
    c = synthetic.CONFIGS[what]
    d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev)
    prob = bench.Problem(torch, engine, synthetic, d, dev)
    ns = d["src_off"][1:] - d["src_off"][:-1]
    nt = d["tgt_off"][1:] - d["tgt_off"][:-1]
    big = torch.maximum(ns, nt)
    mean = int(ns.float().mean())
    bulk = ((5 * mean // 4) + 63) // 64 * 64  # the bulk class bound of icp_launch_host
    for lo, hi, env in ((0, bulk, {"F4L_ICP_THROUGHPUT": "1"}), (bulk, 1 << 30, {"F4L_ICP_THROUGHPUT": "1"})):
        ids = torch.nonzero((big > lo) & (big <= hi), as_tuple=True)[0]
        if ids.numel():
            run_class(d["src"], d["src_off"], d["tgt"], d["tgt_off"], prob.cs, prob.ct, prob.coff, ids, f"({lo}, {hi}]", env)
