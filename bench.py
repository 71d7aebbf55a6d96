#!/usr/bin/env python3
"""bench.py -- M source-points/s through 20-iteration piecewise ICP on a synthetic two-epoch cloud.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1 via torch.distributed.run, one rank per GPU)
prints ONE JSON line on rank 0.

A "step" is one pass of the hot path over one tile, inputs resident in HBM when the clock starts:
    per-patch weighted Kabsch init from the 1-NN correspondences
 -> 20 fixed point-to-point ICP iterations per patch (max_corr_dist 0.1 m, no early exit)
 -> dense displacement rows [s, T s] for every source point
    (all three in ONE launch of f4l_patch_loop; f4l_kabsch_transforms / f4l_piecewise_icp / f4l_apply_transform are the
    same stages as separate calls)
 -> (N > 1) RCCL all-gather of the per-patch results (T, fitness, rmse, iters: 152 B per patch).
Workload at N = 1: BASELINE.json configs[1] ("C2_1M_2k": 1 M points per epoch, 45 x 45 = 2025 patches).
Scaling is weak: tiles are the reference's independent units (<= 1 M points each, configs/landslide/*.yaml
max_pts_per_tile), every rank owns one tile and only the per-patch results are exchanged.

Extra objects on the JSON line:
  roofline     dominant kernel = icp_kernel (the fused loop body); achieved = algorithmic bytes per launch (20 iters x
               24 B + 24 B Kabsch read + 24 B row written = 528 B per source point) / its mean duration
               measured with events on the launch stream; peak = 8 TB/s.
  extras       the same step in float32 mode, with two tiles in flight on alternating streams, and exact kNN-30.
  cpu_baseline the C oracle (oracle/f4l_oracle.c, 1 thread, "port") timed on a bounded sample of the same patches;
               cpu_baseline.all_cores: the same port with its patch loop on all host cores (OpenMP).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
ICP_BYTES_PER_PT_ITER = 24  # 12 B source point + 12 B share of the target patch
FIXED_BYTES_PER_PT = 48      # ... + Kabsch init read (24 B/pt) + displacement row written (24 B/pt): the fused launch does all three
MAX_ITER = 20
MAX_CORR = 0.1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2_1M_2k")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--extras", type=int, default=1, help="also time the float64 parity mode and the kNN-30 kernel (rank 0, N = 1)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from fusion4landslide_amd import engine, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = synthetic.CONFIGS[args.config]
    n, cells, res = cfg["n"], cfg["cells"], cfg["resolution"]
    raster = float(os.environ.get("F4L_BENCH_RASTER", "0")) or None  # experiment: scan-like point order inside patches
    d = synthetic.make_patches(n, cells, res, seed=10 * rank, raster=raster)  # every rank owns its own tile
    P = d["P"]
    src, tgt = torch.from_numpy(d["src"]).to(dev), torch.from_numpy(d["tgt"]).to(dev)
    so, to = torch.from_numpy(d["src_off"]).to(dev), torch.from_numpy(d["tgt_off"]).to(dev)
    eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)

    # Kabsch-init correspondences (inputs of the path, produced upstream by matching in the reference):
    # 1-NN of each source point inside its target patch within 2 x max_corr_dist.  Untimed setup.
    nn, _ = engine.nn_refine(src, so, tgt, to, eye, torch.full((P,), 2 * MAX_CORR, dtype=torch.float64, device=dev),
                             max_tgt_patch=d["max_tgt"], return_rows=False)
    cs_h, ct_h, coff_h = synthetic.correspondences_from_nn(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn.cpu().numpy())
    cs, ct, coff = torch.from_numpy(cs_h).to(dev), torch.from_numpy(ct_h).to(dev), torch.from_numpy(coff_h).to(dev)

    # two sets of receive buffers: the all-gather of step i overlaps the compute of step i + 1 (tiles are independent)
    from fusion4landslide_amd.sharding import TileResultGather
    gather = TileResultGather(dist, torch, world, P, dev) if world > 1 else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i_timed=None, search="f64"):
        # the whole loop body in one launch (f4l_patch_loop): Kabsch init -> ICP -> rows
        if i_timed is not None:
            ev[i_timed][0].record()
        out = engine.patch_loop(src, so, tgt, to, cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=MAX_CORR, max_iter=MAX_ITER,
                                fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], search=search)
        if i_timed is not None:
            ev[i_timed][1].record()
        rows = out["rows"]
        if gather is not None:
            gather.submit(out)
        return out, rows

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out, rows = step(i)
    if gather is not None:
        gather.drain()  # every all-gather of the timed steps has completed before the clock stops
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    icp_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps > 0 else float("nan")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = world * n / (ms_per_step * 1e-3) / 1e6
        alg_bytes = (ICP_BYTES_PER_PT_ITER * MAX_ITER + FIXED_BYTES_PER_PT) * n  # 528 B per source point
        achieved = alg_bytes / (icp_ms * 1e-3) / 1e9
        line = {
            "metric": "M-points/sec piecewise ICP (20 iters, two-epoch cloud)",
            "value": round(value, 3), "unit": "Mpts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.config, "points_per_epoch_per_gpu": n, "patches_per_gpu": P,
                       "icp": "point2point, 20 fixed iters, max_corr_dist 0.1 m, float64 search (parity mode)", "parallelism": f"tiles x{world}",
                       "mean_fitness": round(float(out["fitness"].mean().item()), 4)},
            "roofline": {"bound": "hbm", "kernel": "icp_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(icp_ms, 4), "algorithmic_bytes": alg_bytes},
        }
        # HBM bytes per launch of icp_kernel from the PMC counters: cannot be collected from inside this process; they
        # come from the committed rocprofv3 passes of this same command (profiles/README.md), when the workload matches
        tr = os.path.join(ROOT, "profiles", "icp_kernel_traffic.json")
        if os.path.exists(tr):
            t = json.load(open(tr))
            if t.get("workload") == args.config:
                line["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = t["source"]
        if world == 1:
            # the box's own device-to-device copy rate (SURVEY.md 8d asks for the fraction of nominal AND of measured)
            copy_gbs = measured_copy_gbs(torch, dev)
            line["roofline"]["peak_copy_measured"] = round(copy_gbs, 1)
            line["roofline"]["frac_of_copy_measured"] = round(achieved / copy_gbs, 5)
        if world == 1 and args.extras:
            line["extras"] = extras(torch, engine, step, src, args)
        if world == 1 and args.cpu_seconds > 0:  # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(d, cs_h, ct_h, coff_h, args.cpu_seconds)
            if args.extras:
                line["cpu_baseline_supervoxel"] = cpu_baseline_supervoxel(torch, engine, d)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measured_copy_gbs(torch, dev, gib=1.0, reps=10):
    """Read + write bytes per second of a large device-to-device copy (the practical HBM peak of this box)."""
    n = int(gib * (1 << 30)) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def extras(torch, engine, step, src, args):
    """Secondary figures of SURVEY.md 8(d), outside the timed region of the headline metric."""
    out = {}
    n = src.shape[0]

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # the same step in the fast mode (float32 search and partial sums; not the headline: an ill-posed patch can end in
    # a different local solution than the float64 arithmetic of the reference)
    s32 = timed(lambda: step(search="f32"), max(2, args.steps // 2))
    out["fast_mode_f32"] = {"value": round(n / s32 / 1e6, 3), "unit": "Mpts/s", "ms_per_step": round(1e3 * s32, 4)}
    # consecutive tiles are independent: issued on two alternating streams, the draining tail of one launch (a tile is
    # only two rounds of workgroups) overlaps the head of the next.  Same step, same work per tile; not the headline,
    # because the per-launch roofline above is defined on an undisturbed launch.
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    reps = max(4, 2 * args.steps)

    def alternate():
        for i in range(reps):
            with torch.cuda.stream(streams[i % 2]):
                step()
    alternate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    alternate()
    torch.cuda.synchronize()
    s2 = (time.perf_counter() - t0) / reps
    out["two_tiles_in_flight"] = {"value": round(n / s2 / 1e6, 3), "unit": "Mpts/s", "ms_per_tile": round(1e3 * s2, 4),
                                  "note": "float64 mode, tiles issued on two alternating HIP streams"}
    # exact kNN-30 of the source epoch (supervoxel stage): 12 B read + 120 B written per point
    sk = timed(lambda: engine.knn(src, 30), 3)
    out["knn30"] = {"value": round(n / sk / 1e6, 3), "unit": "Mpts/s", "ms": round(1e3 * sk, 3),
                    "achieved_GBs": round(132.0 * n / sk / 1e9, 2), "frac_of_hbm_peak": round(132.0 * n / sk / 1e9 / HBM_PEAK_GBS, 5),
                    "note": "f4l_knn end to end (binning + sort + search), algorithmic 132 B/pt"}
    return out


def cpu_baseline_supervoxel(torch, engine, d, n=200_000, k=30):
    """SURVEY.md 8(a) row a1 on a bounded sample (the first n points of the source epoch): `computeSupervoxel` without its
    file I/O.  CPU side: the reference's OWN templates when oracle/_ref/libf4l_ref.so is there (kind "reference": the
    header-only codelibrary driven by oracle/ref_harness.cpp), else the C restatement (kind "port"); one thread, like
    the reference.  GPU side: f4l_supervoxel (kNN + normals on the device, the order-dependent segmentation on the
    host), labels compared for identity."""
    import numpy as np

    from oracle import oracle as O

    xyz = np.ascontiguousarray(d["meta"]["src"][:n])
    res = 1.386
    kind = "reference" if O.have_ref() else "port"
    t = time.perf_counter()
    ref = O.ref_supervoxel(xyz, k, res) if kind == "reference" else O.supervoxel(xyz, k, res)
    cpu_s = time.perf_counter() - t
    dev_xyz = torch.from_numpy(xyz).cuda()
    engine.supervoxel(dev_xyz, k, res)  # warm-up
    torch.cuda.synchronize()
    t = time.perf_counter()
    labels, K = engine.supervoxel(dev_xyz, k, res)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t
    same = bool(np.array_equal(labels.cpu().numpy(), ref["labels"])) and K == ref["n_supervoxels"]
    return {"value": round(len(xyz) / cpu_s / 1e6, 4), "unit": "Mpts/s", "cores": 1, "kind": kind,
            "sample": f"first {len(xyz)} source points, k={k}, resolution {res} m, {cpu_s:.1f} s",
            "this_repo": {"value": round(len(xyz) / gpu_s / 1e6, 4), "unit": "Mpts/s", "seconds": round(gpu_s, 3),
                          "labels_identical": same, "n_supervoxels": int(K),
                          "note": "kNN + normals on the GPU (milliseconds), segmentation replayed on one host core"}}


def cpu_baseline(d, cs, ct, coff, budget_s):
    """The oracle (single-thread C port of the same step) on a bounded prefix of the tile's patches."""
    import numpy as np

    from oracle import oracle as O

    P = d["P"]
    # calibrate on a few patches, then size the sample to the budget
    def run(p_lo, p_hi):
        s0, s1, t0, t1 = d["src_off"][p_lo], d["src_off"][p_hi], d["tgt_off"][p_lo], d["tgt_off"][p_hi]
        so, to = d["src_off"][p_lo:p_hi + 1] - s0, d["tgt_off"][p_lo:p_hi + 1] - t0
        co = coff[p_lo:p_hi + 1] - coff[p_lo]
        c0, c1 = coff[p_lo], coff[p_hi]
        t = time.perf_counter()
        R, tt = O.kabsch_batched(cs[c0:c1], ct[c0:c1], co, eps=1e-6)
        T0 = np.tile(np.eye(4), (p_hi - p_lo, 1, 1))
        T0[:, :3, :3] = R
        T0[:, :3, 3] = tt
        res = O.piecewise_icp(d["src"][s0:s1], so, d["tgt"][t0:t1], to, init_T=T0, max_corr_dist=MAX_CORR,
                              max_iter=MAX_ITER, fixed_iters=True)
        s = d["src"][s0:s1].astype(np.float64)
        pid = np.repeat(np.arange(p_hi - p_lo), np.diff(so))
        _ = np.einsum("nij,nj->ni", res["T"][pid, :3, :3], s) + res["T"][pid, :3, 3]
        return time.perf_counter() - t, int(s1 - s0)

    probe = min(P, 16)
    dt, npts = run(0, probe)
    rate = npts / dt
    want = int(min(P, max(probe, budget_s * rate / (d["src"].shape[0] / P))))
    dt, npts = run(0, want)
    out = {"value": round(npts / dt / 1e6, 5), "unit": "Mpts/s", "cores": 1, "kind": "port",
           "sample": f"first {want} of {P} patches ({npts} source points), same Kabsch+ICP(20)+apply step, {dt:.1f} s"}
    # the same port with its patch loop on every core this process may use (OpenMP)
    cores = host_cores()
    if cores > 1:
        used = O.set_threads(cores)
        try:
            want_all = int(min(P, max(probe, want * used * 0.5)))
            run(0, min(want_all, 4 * used))  # thread start-up outside the clock
            dt_all, npts_all, reps = 0.0, 0, 0
            while dt_all < 0.3 * budget_s and reps < 20:
                dt1, n1 = run(0, want_all)
                dt_all, npts_all, reps = dt_all + dt1, npts_all + n1, reps + 1
        finally:
            O.set_threads(1)
        out["all_cores"] = {"value": round(npts_all / dt_all / 1e6, 5), "unit": "Mpts/s", "cores": used,
                            "sample": f"first {want_all} of {P} patches x{reps} ({npts_all} source points), {dt_all:.1f} s"}
    return out


def host_cores():
    """CPUs this process can really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    cores = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


if __name__ == "__main__":
    main()
