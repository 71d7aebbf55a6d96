#!/usr/bin/env python3
"""bench.py -- M source-points/s through 20-iteration piecewise ICP on ONE synthetic two-epoch cloud.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
  N > 1: either launched by the driver through torch.distributed.run (WORLD_SIZE = N in the environment), or -- when
  WORLD_SIZE is unset -- this script starts N ranks itself (a `python -m torch.distributed.run` child process, before
  anything in this process touches the GPU) and relays rank 0's line.  WORLD_SIZE != N is an error (exit code 2).

Workload at every N: BASELINE.json's metric config "C4_50M_100k" (50 M points per epoch, 316 x 316 = 99 856 patches), which
fits one MI355X (2.4 GB of inputs and outputs out of 288 GB).  The cloud comes from a counter-based generator ON THE DEVICE
(fusion4landslide_amd.synthetic: any rank can produce any part of it); its patches are assigned to the ranks by LPT on
|src| x |tgt| (fusion4landslide_amd.sharding) and every rank generates and keeps only its own share
(synthetic.make_rank_share_device): total work is fixed as N grows ("scaling": "strong"), `value` = the cloud's 50 M source
points / max-over-ranks time.

A "step" is one pass of the hot path over the rank's share, inputs resident in HBM when the clock starts:
    per-patch weighted Kabsch init from the 1-NN correspondences
 -> 20 fixed point-to-point ICP iterations per patch (max_corr_dist 0.1 m, no early exit, float64 = Open3D's arithmetic)
 -> dense displacement rows [s, T s] for every source point          (all three in ONE launch of f4l_patch_loop)
 -> all-gather of the per-patch results (T, fitness, rmse, iters: 152 B per patch) over RCCL; the only exchange step.

Extra objects on the JSON line:
  roofline       dominant kernel = icp_kernel (the fused loop body) on rank 0: achieved = algorithmic bytes per launch
                 (20 iters x 24 B + 24 B Kabsch read + 24 B row written = 528 B per source point) / its mean
                 duration measured with events on the launch stream; peak = 8 TB/s; traffic = HBM bytes per launch from
                 the committed rocprofv3 --pmc passes of this command (null when the kernel source changed since).
                 `bound` and the `valu` sub-object (vector-issue cycles per launch against 256 CUs x 4 SIMDs x 2.4 GHz) come from
                 the same committed passes: the kernel is LDS resident, HBM is not what bounds it.
  roofline_knn   the same for the exact kNN-30 kernel of the supervoxel stage (132 B per point), the kernel north_star
                 asks HBM numbers for (N = 1 only).
  roofline_supervoxel  the same for the whole partition stage as the path runs it by default: f4l_supervoxel (the reference's
                 labels) on 10 M points, 160 B per point; `variant_parallel` inside it: the opt-in f4l_supervoxel_parallel.
  extras         the other single-GPU BASELINE configs (C2 1 M, C3 10 M dense) and the float32 fast mode (N = 1 only).
  cpu_baseline   the C oracle (oracle/f4l_oracle.c, "port") timed on a bounded prefix of the same patches, 1 thread, and
                 with its patch loop on all host cores (rank 0, N = 1 only).

--dry-run: orchestration check without a GPU (gloo, CPU tensors, a stub instead of the kernel launch; `value` is null).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# (RCCL / tensor sharing between the ranks' processes needs dmabuf IPC on this pool: the box exports it already; a rank started
#  from a shell that lost it must not fail with `hipIpcGetMemHandle: invalid argument` -- set before any HIP call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
ICP_BYTES_PER_PT_ITER = 24  # 12 B source point + 12 B share of the target patch
FIXED_BYTES_PER_PT = 48      # ... + Kabsch init read (24 B/pt) + displacement row written (24 B/pt): the fused launch does all three
KNN_BYTES_PER_PT = 132       # 12 B read + 30 x 4 B neighbour indices written
MAX_ITER = 20
MAX_CORR = 0.1
SV_BYTES_PER_PT = 160        # kNN 132 B + normals 24 B + labels 4 B per point
VALU_PEAK_GCYC = 256 * 4 * 2.4  # G SIMD-cycles of vector issue per second: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md); a wave64
                                # float32 / integer instruction takes 2 of them, a float64 one 4
KERNEL_SOURCES = {"icp": ("icp.hip", "icp_rows.h", "ldlt6.h", "patch_grid.h", "f4l_device.h"), "knn": ("knn.hip", "lane_topk.h", "topk.h", "f4l_device.h"),
                  "supervoxel": ("supervoxel_gpu.hip", "sv_metric.h", "select.hip", "knn.hip", "lane_topk.h", "topk.h", "f4l_device.h"),
                  "supervoxel_exact": ("supervoxel_exact.hip", "sv_metric.h", "select.hip", "knn.hip", "lane_topk.h", "topk.h", "f4l_device.h")}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C4_50M_100k")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--extras", type=int, default=1, help="also time C2, C3, the float32 mode and the kNN-30 kernel (N = 1)")
    ap.add_argument("--dry-run", action="store_true", help="orchestration only: gloo + CPU tensors + a stub launch")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`--gpus N` without a launcher around us: start the N ranks as a child torch.distributed.run (this process has not
    touched the GPU and never will) and relay what they print."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(world_env or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start one rank per GPU "
              f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)", file=sys.stderr)
        raise SystemExit(2)
    run_rank(args, world)


def run_rank(args, world):
    import numpy as np
    import torch
    import torch.distributed as dist

    from fusion4landslide_amd import sharding, synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        engine = None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists for the product path; --dry-run checks the orchestration)")
        from fusion4landslide_amd import engine
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    cfg = synthetic.CONFIGS[args.config]
    n, cells, res = cfg["n"], cfg["cells"], cfg["resolution"]
    t_setup = time.perf_counter()
    setup_parts = {}
    if not dry:  # the first device call of the process: context, code objects of torch and of libf4l_hip.so
        torch.zeros(1, device=dev).add_(1)
        engine.lib()
        sync()
    setup_parts["device_and_library_start"] = time.perf_counter() - t_setup
    if world == 1:
        cloud = synthetic.make_patches_device(n, cells, res, dev, seed=0)
        P_total = cloud["P"]
        d, ids_per_rank = sharding.shard_cloud(cloud, rank, world)
        del cloud
    else:
        # every rank produces ITS share of the one cloud straight from the counter-based generator (a counting pass for the LPT
        # assignment, a keeping pass): no rank ever holds the whole 50 M-point cloud, and nothing is scattered between ranks
        # (the counting pass is shared between the ranks: every world-th chunk each, one all_reduce of the counts)
        d, ids_per_rank = synthetic.make_rank_share_device(n, cells, res, dev, rank, world, seed=0, dist=dist)
        P_total = cells * cells
    if not dry:
        torch.cuda.empty_cache()
    sync()
    setup_parts["generate_cloud"] = time.perf_counter() - t_setup - setup_parts["device_and_library_start"]
    P, n_mine = d["P"], d["n_src"]
    mine = torch.from_numpy(ids_per_rank[rank]).to(dev)
    prob = Problem(torch, engine, synthetic, d, dev, dry, mine)
    gather = sharding.PatchResultGather(dist, torch, ids_per_rank, rank, dev)
    sync()
    t_setup = time.perf_counter() - t_setup
    setup_parts["point_matches"] = t_setup - setup_parts["generate_cloud"] - setup_parts["device_and_library_start"]

    ev = [] if dry else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i_timed=None):
        if i_timed is not None and ev:
            ev[i_timed][0].record()
        out = prob.step()
        if i_timed is not None and ev:
            ev[i_timed][1].record()
        gather.submit(out)  # async all-gather; overlaps the next step's launch
        return out

    for _ in range(args.warmup):
        step()
    gather.drain()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    gather.drain()  # every all-gather of the timed steps has completed before the clock stops
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t_local = time.perf_counter() - t0   # (every rank's own clock over the same barrier-to-barrier region)
    elapsed = t_local
    per_rank = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    full = gather.latest() if args.steps + args.warmup > 0 else None
    if world > 1:
        # what the one number above hides: every rank's own compute time per step (events on its launch stream around
        # its f4l_patch_loop: the LPT balance), its points and patches, and the all-gather by itself -- timed AFTER the metric's
        # region, un-overlapped: submit + drain between two synchronisations, mean of a few
        own_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if (ev and args.steps > 0) else 0.0
        ag = []
        for _ in range(5 if (args.steps + args.warmup > 0) else 0):
            sync(); dist.barrier(); sync()
            ta = time.perf_counter()
            gather.submit(out); gather.drain(); sync()
            ag.append(1e3 * (time.perf_counter() - ta))
        mine_stats = torch.tensor([own_ms, float(n_mine), float(P), float(np.mean(ag)) if ag else 0.0, 1e3 * t_local / max(args.steps, 1)],
                                  dtype=torch.float64, device=dev)
        allr = torch.zeros((world, 5), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine_stats.view(1, 5))
        allr = allr.cpu().numpy()
        mm = lambda c: {"min": round(float(allr[:, c].min()), 4), "max": round(float(allr[:, c].max()), 4),  # noqa: E731
                        "mean": round(float(allr[:, c].mean()), 4)}
        per_rank = {"step_ms": mm(0), "points": mm(1), "patches": mm(2), "allgather_ms": mm(3), "wall_ms_per_step": mm(4),
                    "allgather_bytes_per_rank": int(gather.pmax * 19 * 8),
                    "note": "step_ms = a rank's own f4l_patch_loop per step (events; 0 in a dry run); allgather_ms = one all-gather of the "
                            "per-patch results alone, after the timed region"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        line = {
            "metric": "M-points/sec piecewise ICP (20 iters, two-epoch cloud)",
            "value": None if dry else round(n / (ms_per_step * 1e-3) / 1e6, 3), "unit": "Mpts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "timed_region_s": round(elapsed, 4),
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.config, "points_per_epoch": n, "patches": P_total,
                       "points_on_rank0": n_mine, "patches_on_rank0": P,
                       "icp": "point2point, 20 fixed iters, max_corr_dist 0.1 m, float64 search (parity mode)",
                       "parallelism": f"one cloud, patches LPT-sharded x{world}, all-gather of per-patch results",
                       # what the process group really is (the first multi-GPU run proves "RCCL saw N ranks" from this line alone)
                       "dist_backend": dist.get_backend() if world > 1 else None,
                       "dist_world_size": dist.get_world_size() if world > 1 else 1,
                       "setup_seconds": round(t_setup, 2),
                       # (on a fresh box most of it is the first device call -- context, code objects paged in from a cold image --
                       #  and the allocator's first gigabytes; untimed, outside the metric)
                       "setup_breakdown_s": {k: round(v, 2) for k, v in setup_parts.items()}},
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
        if dry:
            line["dry_run"] = True
            line["dry_run_gather_ok"] = bool(full is not None and torch.equal(
                full["fitness"], torch.arange(P_total, dtype=torch.float64)) and full["T"].shape == (P_total, 4, 4))
        else:
            line["config"]["mean_fitness"] = round(float(full["fitness"].mean().item()), 4)
            icp_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps > 0 else float("nan")
            alg_bytes = (ICP_BYTES_PER_PT_ITER * MAX_ITER + FIXED_BYTES_PER_PT) * n_mine  # 528 B per source point of the launch
            achieved = alg_bytes / (icp_ms * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "icp_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                                "kernel_ms": round(icp_ms, 4), "algorithmic_bytes": alg_bytes,
                                "note": "achieved = algorithmic bytes / kernel time (contract: events on the launch stream around one "
                                        "f4l_patch_loop = the icp_kernel launch of the bulk of the patches with the few large border patches' "
                                        "launch beside it); `bound` is what the committed counter passes show (profiles/README.md): the patch "
                                        "pair is LDS resident, the measured HBM traffic is a fraction of the algorithmic bytes, and the "
                                        "vector-issue share under `valu` is the larger of the two"}
            attach_counters(line["roofline"], "icp_kernel_counters.json", "icp", args.config if world == 1 else None, icp_ms)
            if world == 1:
                copy_gbs = measured_copy_gbs(torch, dev)
                line["roofline"]["peak_copy_measured"] = round(copy_gbs, 1)
                line["roofline"]["frac_of_copy_measured"] = round(achieved / copy_gbs, 5)
                if args.extras:
                    line["roofline_knn"] = knn_roofline(torch, engine, d["src"][:10_000_000])
                    line["roofline_supervoxel"] = supervoxel_roofline(torch, engine, d["src"][:10_000_000])
                    line["extras"] = extras(torch, engine, synthetic, prob, dev, args)
                if args.cpu_seconds > 0:  # rank 0 at N = 1 only
                    line["cpu_baseline"] = cpu_baseline(prob, args.cpu_seconds)
                    if args.extras:
                        line["cpu_baseline_supervoxel"] = cpu_baseline_supervoxel(torch, engine, d["src"])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Problem:
    """One rank's share of a cloud, resident on the device, with the Kabsch-init correspondences of the path (inputs
    produced upstream by matching in the reference; here the 1-NN of each source point inside its target patch within
    2 x max_corr_dist -- untimed set-up)."""

    def __init__(self, torch, engine, synthetic, d, dev, dry=False, ids=None):
        self.torch, self.engine, self.d, self.dry, self.ids = torch, engine, d, dry, ids
        P = d["P"]
        if dry:
            self.cs = self.ct = torch.zeros((0, 3), dtype=torch.float32)
            self.coff = torch.zeros(P + 1, dtype=torch.int64)
            return
        eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
        nn, _ = engine.nn_refine(d["src"], d["src_off"], d["tgt"], d["tgt_off"], eye,
                                 torch.full((P,), 2 * MAX_CORR, dtype=torch.float64, device=dev),
                                 max_tgt_patch=d["max_tgt"], return_rows=False)
        self.cs, self.ct, self.coff = synthetic.correspondences_from_nn_device(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn)

    def step(self, search="f64", icp_type="point2point", tgt_normals=None):
        d, torch = self.d, self.torch
        if self.dry:  # stub launch: identity transforms, fitness = GLOBAL patch id (checked after the gather)
            P = d["P"]
            return dict(T=torch.eye(4, dtype=torch.float64).repeat(P, 1, 1), fitness=self.ids.to(torch.float64),
                        rmse=torch.zeros(P, dtype=torch.float64), iters=torch.full((P,), MAX_ITER, dtype=torch.int32))
        # the whole loop body in one launch (f4l_patch_loop): Kabsch init -> ICP -> rows
        return self.engine.patch_loop(d["src"], d["src_off"], d["tgt"], d["tgt_off"], self.cs, self.ct, self.coff, None, 0.0, 1e-6,
                                      max_corr_dist=MAX_CORR, max_iter=MAX_ITER, fixed_iters=True, max_src_patch=d["max_src"],
                                      max_tgt_patch=d["max_tgt"], search=search, icp_type=icp_type, tgt_normals=tgt_normals)

    def host_prefix(self, p_hi):
        """The first p_hi patches as host numpy arrays (for the CPU baseline: the same bits the GPU worked on)."""
        d = self.d
        s1, t1, c1 = int(d["src_off"][p_hi]), int(d["tgt_off"][p_hi]), int(self.coff[p_hi])
        h = lambda t: t.cpu().numpy()  # noqa: E731
        return dict(src=h(d["src"][:s1]), src_off=h(d["src_off"][:p_hi + 1]), tgt=h(d["tgt"][:t1]), tgt_off=h(d["tgt_off"][:p_hi + 1]),
                    cs=h(self.cs[:c1]), ct=h(self.ct[:c1]), coff=h(self.coff[:p_hi + 1]))


def kernel_source_hash(which):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[which]:
        h.update(open(os.path.join(ROOT, "fusion4landslide_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def attach_counters(roof, fname, which, workload, kernel_ms, units=None):
    """HBM bytes and instruction counts cannot be collected from inside this process; they come from the committed rocprofv3
    --pmc passes of the same call (profiles/README.md, tools/gpu/pmc_passes.py -> tools/make_roofline_profiles.py).  Used only when
    the workload AND the kernel sources are the ones the passes were taken on -- otherwise `traffic` stays null rather than going
    stale.  `units`: the file's figures are per `units_in_profile` points; this run processed `units` (same code path, other size)."""
    path = os.path.join(ROOT, "profiles", fname)
    if workload is None or not os.path.exists(path):
        return
    t = json.load(open(path))
    if t.get("workload") != workload:
        return
    if t.get("kernel_source_sha256_16") != kernel_source_hash(which):
        roof["traffic_note"] = "profiles/%s was taken on other kernel sources; not reported" % fname
        return
    scale = 1.0 if units is None else units / float(t["units_in_profile"])
    roof["traffic"] = int(t["hbm_bytes_per_launch"] * scale)
    roof["traffic_source"] = t["source"]
    roof["achieved_measured_traffic"] = round(roof["traffic"] / (kernel_ms * 1e-3) / 1e9, 2)
    roof["frac_measured_traffic"] = round(roof["achieved_measured_traffic"] / HBM_PEAK_GBS, 5)
    v = t.get("valu")
    if v:
        cyc = v["issue_cycles_per_launch"] * scale
        ach = cyc / (kernel_ms * 1e-3) / 1e9
        roof["valu"] = {"achieved": round(ach, 1), "peak": VALU_PEAK_GCYC, "unit": "G SIMD-cycles/s of vector issue", "frac": round(ach / VALU_PEAK_GCYC, 4),
                        "wave_insts_per_launch": int(v["insts_per_launch"] * scale), "f64_share": v["f64_share"], "weights": v["weights"]}
        roof["bound"] = "valu" if roof["valu"]["frac"] > roof["frac_measured_traffic"] else "hbm"


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


def _timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def knn_roofline(torch, engine, xyz, k=30):
    """Exact kNN-30 of (up to 10 M points of) the source epoch, the search stage of `computeSupervoxel`: f4l_knn end to end
    (binning + sort + search kernel) timed with events on the launch stream; 12 B read + 120 B written per point."""
    n = xyz.shape[0]
    s = _timed(torch, lambda: engine.knn(xyz, k), 5)
    ach = KNN_BYTES_PER_PT * n / s / 1e9
    roof = {"bound": "hbm", "kernel": "knn_lanes_kernel (f4l_knn end to end: binning + search)", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "kernel_ms": round(1e3 * s, 4),
            "algorithmic_bytes": KNN_BYTES_PER_PT * n, "points": n, "k": k, "Mpts_per_s": round(n / s / 1e6, 2)}
    attach_counters(roof, "knn_counters.json", "knn", "f4l_knn k=30", 1e3 * s, units=n)
    return roof


def supervoxel_roofline(torch, engine, xyz, k=30):
    """The partition stage of the full path on (up to 10 M points of) the source epoch, as the path and both entry points run it by
    default: f4l_supervoxel end to end -- exact kNN-30 + PCA normals + the REFERENCE's segmentation on the device
    (csrc/supervoxel_exact.hip) -- at the path's own resolution rule (sqrt(3) x 10 x median point spacing,
    src/coarse_to_fine_matching_base.py:2668-2671), timed with events on the launch stream; SURVEY.md 8(d) prices the stage at
    132 + 24 + 4 = 160 B per point.  `variant_parallel`: the same for the opt-in f4l_supervoxel_parallel (other labels)."""
    import numpy as np
    n = xyz.shape[0]
    res = float(np.sqrt(3.0) * 10.0 * engine.median_resolution(xyz))
    K = [0]

    def one(fn, which, fname, workload, what, note, reps):
        def run():
            K[0] = fn(xyz, k, res)[1]
        s = _timed(torch, run, reps)
        ach = SV_BYTES_PER_PT * n / s / 1e9
        roof = {"bound": "hbm", "kernel": what, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": None, "kernel_ms": round(1e3 * s, 4), "algorithmic_bytes": SV_BYTES_PER_PT * n, "points": n, "k": k,
                "resolution_m": round(res, 4), "supervoxels": int(K[0]), "Mpts_per_s": round(n / s / 1e6, 2), "note": note}
        attach_counters(roof, fname, which, workload, 1e3 * s, units=n)
        return roof
    roof = one(engine.supervoxel, "supervoxel_exact", "supervoxel_exact_counters.json", "f4l_supervoxel k=30",
               "f4l_supervoxel (knn_lanes_kernel + svx::eval_kernel / xch_eval_kernel passes of the reference's segmentation)",
               "the reference's sequential fusion and FIFO exchange as fixed points of parallel passes: every pass re-evaluates the "
               "representatives whose inputs changed, through dependent gathers (list -> root -> claim -> record -> metric); bound by those "
               "and by the number of passes, not by HBM", 2)
    roof["variant_parallel"] = one(engine.supervoxel_parallel, "supervoxel", "supervoxel_counters.json", "f4l_supervoxel_parallel k=30",
                                   "f4l_supervoxel_parallel (knn_lanes_kernel + the svg:: kernels of the segmentation, ~270 launches)",
                                   "a graph contraction: ~14 passes over edge lists that shrink from 30 edges per point; its passes wait for "
                                   "dependent scattered loads (profiles/README.md)", 3)
    return roof


def extras(torch, engine, synthetic, prob, dev, args):
    """Secondary figures of SURVEY.md 8(d), outside the timed region of the headline metric: the same step on the other
    single-GPU configs of BASELINE.json, and in the float32 fast mode."""
    out = {}
    n = prob.d["n_src"]
    # the fast mode (float32 search and partial sums; not the headline: an ill-posed patch can end in a different local
    # solution than the float64 arithmetic of the reference)
    s32 = _timed(torch, lambda: prob.step(search="f32"), max(2, args.steps // 5))
    out["fast_mode_f32"] = {"workload": args.config, "value": round(n / s32 / 1e6, 3), "unit": "Mpts/s", "ms_per_step": round(1e3 * s32, 4)}
    # the other estimator of utils/o3d_tools.py:46-50: point-to-plane, with the target normals `estimate_normals()` gives every
    # patch cloud (:29-30, f4l_patch_normals) -- the same step on the same cloud
    def normals():
        return engine.patch_normals(prob.d["tgt"], prob.d["tgt_off"], 30, max_patch=prob.d["max_tgt"], f64=True)  # (doubles, like Open3D's)
    normals()
    s_n = _timed(torch, normals, 2)
    nrm = normals()
    s_pl = _timed(torch, lambda: prob.step(icp_type="point2plane", tgt_normals=nrm), max(2, args.steps // 10))
    out["point2plane"] = {"workload": args.config, "value": round(n / s_pl / 1e6, 3), "unit": "Mpts/s", "ms_per_step": round(1e3 * s_pl, 4),
                          "estimate_normals_ms": round(1e3 * s_n, 3)}
    del nrm
    for name in ("C1_50k_64", "C2_1M_2k", "C2x16_16M_32k", "C3_10M_20k"):  # (C2x16: the C2 density at 32 400 patches, one rank's regime of an 8-GPU run)
        if name == args.config:
            continue
        c = synthetic.CONFIGS[name]
        d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=0)
        d["n_src"] = c["n"]
        p = Problem(torch, engine, synthetic, d, dev)
        s = _timed(torch, p.step, 10)
        out[name] = {"value": round(c["n"] / s / 1e6, 3), "unit": "Mpts/s", "ms_per_step": round(1e3 * s, 4), "patches": d["P"],
                     "largest_patch": max(d["max_src"], d["max_tgt"]),
                     "algorithmic_GBs": round(528.0 * c["n"] / s / 1e9, 1), "frac_of_hbm_peak": round(528.0 * c["n"] / s / 1e9 / HBM_PEAK_GBS, 5)}
        del p, d
        torch.cuda.empty_cache()
    # the reference's real unit of work is a tile of <= 1 M points (main_fusion.py:134, fusion_brienz.yaml:25-26): eight C2 tiles,
    # tile by tile, and as ONE launch over their concatenated patches (engine.patch_loop_tiles; the concatenation is inside the clock)
    c = synthetic.CONFIGS["C2_1M_2k"]
    tiles = []
    for seed in range(8):
        d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=seed)
        p = Problem(torch, engine, synthetic, d, dev)
        tiles.append(dict(src=d["src"], src_off=d["src_off"], tgt=d["tgt"], tgt_off=d["tgt_off"], corr_src=p.cs, corr_ref=p.ct, corr_off=p.coff,
                          max_src=d["max_src"], max_tgt=d["max_tgt"]))
    kw = dict(max_corr_dist=MAX_CORR, max_iter=MAX_ITER, fixed_iters=True)

    def one_by_one():
        for t in tiles:
            engine.patch_loop(t["src"], t["src_off"], t["tgt"], t["tgt_off"], t["corr_src"], t["corr_ref"], t["corr_off"],
                              max_src_patch=t["max_src"], max_tgt_patch=t["max_tgt"], **kw)
    s_each = _timed(torch, one_by_one, 5)
    s_all = _timed(torch, lambda: engine.patch_loop_tiles(tiles, **kw), 5)
    m = engine.merge_tiles(tiles)
    m.pop("split")
    margs = [m.pop(k) for k in ("src", "src_off", "tgt", "tgt_off", "corr_src", "corr_ref", "corr_off", "corr_weights")]
    s_merged = _timed(torch, lambda: engine.patch_loop(*margs, **m, **kw), 5)
    out["C2x8_tiles"] = {"workload": "8 tiles of C2_1M_2k (8 M points, 16 200 patches), float64 mode", "unit": "Mpts/s",
                         "tile_by_tile": round(8 * c["n"] / s_each / 1e6, 3), "ms_tile_by_tile": round(1e3 * s_each, 4),
                         "one_launch": round(8 * c["n"] / s_merged / 1e6, 3), "ms_one_launch": round(1e3 * s_merged, 4),
                         "one_launch_incl_concatenation": round(8 * c["n"] / s_all / 1e6, 3), "ms_incl_concatenation": round(1e3 * s_all, 4),
                         "note": "one_launch: the tiles' patch arrays already in one buffer (engine.merge_tiles once, then engine.patch_loop); "
                                 "incl_concatenation: engine.patch_loop_tiles on separate per-tile arrays, the copies inside the clock"}
    del tiles, m, margs
    torch.cuda.empty_cache()
    # configs[4] on one GPU: the whole hot path of a tile -- supervoxel partition (all on the device), patches, point matches,
    # Kabsch + 20-iteration ICP + rows, nearest-neighbour refinement -- end to end, stage by stage
    from fusion4landslide_amd import pipeline
    for n_pts, cells in ((1_000_000, 45), (10_000_000, 142), (100_000_000, 450)):  # (the last one: BASELINE.json configs[4], C5_100M_full)
        c = synthetic.make_patches_device(n_pts, cells, 1.386, dev, seed=0)
        src, tgt = c["src"], c["tgt"]
        del c
        for part, note in (("identical", "the reference's labels (f4l_supervoxel: the default of pipeline.full_path and of both entry points)"),
                           ("parallel", "parallel variant (f4l_partition_neighbours + f4l_partition_segment): the reference's K and criteria, other labels")):
            pipeline.full_path(src, tgt, max_iter=MAX_ITER, fixed_iters=True, partition=part)  # warm-up
            r = pipeline.full_path(src, tgt, max_iter=MAX_ITER, fixed_iters=True, partition=part)
            out[f"full_path_{n_pts // 1_000_000}M_{part}_partition"] = {
                "value": round(n_pts / r["stage_ms"]["total"] / 1e3, 3), "unit": "Mpts/s", "supervoxels": int(r["K"]), "resolution_m": round(r["resolution"], 4),
                "stage_ms": {k: round(v, 3) for k, v in r["stage_ms"].items()}, "mean_fitness": round(float(r["fitness"].mean().item()), 4), "partition": note}
            del r
            engine.release_scratch()
            torch.cuda.empty_cache()
        if n_pts == 1_000_000:
            # The reference's labels depend on the ORDER of the points (the sequential fusion visits them by index), and so does the
            # number of passes the fixed-point form needs.  The clouds above are patch by patch (random inside a patch): the
            # favourable case.  The same cloud in the order a voxel-grid filter or the tiler writes, in Morton order and shuffled:
            res_m = float((3.0 ** 0.5) * 10.0 * engine.median_resolution(src))
            cell = ((src - src.min(0).values) / 0.05).floor().to(torch.int64)
            nx, ny = int(cell[:, 0].max()) + 1, int(cell[:, 1].max()) + 1

            def spread(v):
                v = v & 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F
                v = (v | (v << 2)) & 0x33333333; return (v | (v << 1)) & 0x55555555
            orders = {"patch_by_patch": None,
                      "voxel_index_x_fastest": torch.argsort((cell[:, 2] * ny + cell[:, 1]) * nx + cell[:, 0], stable=True),
                      "morton_xy": torch.argsort(spread(cell[:, 0]) | (spread(cell[:, 1]) << 1), stable=True),
                      "shuffled": torch.randperm(n_pts, device=dev, generator=torch.Generator(device=dev).manual_seed(1))}
            ms = {}
            for name, o in orders.items():
                pts = src if o is None else src[o].contiguous()
                engine.supervoxel(pts, 30, res_m)
                ms[name] = round(1e3 * _timed(torch, lambda: engine.supervoxel(pts, 30, res_m), 3), 2)
                del pts
            out["partition_1M_by_point_order_ms"] = dict(ms, note="f4l_supervoxel (the reference's labels) of the same 1 M points in four orders: the "
                                                         "sequential algorithm's dependency chains, hence the passes, follow the index order")
            del cell, orders
        del src, tgt
        torch.cuda.empty_cache()
    return out


def cpu_baseline_supervoxel(torch, engine, src, n=200_000, k=30):
    """SURVEY.md 8(a) row a1 on a bounded sample (the first n points of the source epoch: patch-contiguous, i.e. a compact
    strip of the cloud): `computeSupervoxel` without its file I/O.  CPU side: the C restatement (oracle/f4l_oracle.c,
    label-identical to the reference's templates on the golden clouds; kind "port"), one thread like the reference.
    The reference's own templates are timed in the build container only: nothing built from
    /root/reference runs on the GPU box."""
    import numpy as np

    from oracle import oracle as O

    xyz = np.ascontiguousarray(src[:n].cpu().numpy())
    res = 1.386
    t = time.perf_counter()
    ref = O.supervoxel(xyz, k, res)
    cpu_s = time.perf_counter() - t
    dev_xyz = torch.from_numpy(xyz).cuda()
    out = {"value": round(len(xyz) / cpu_s / 1e6, 4), "unit": "Mpts/s", "cores": 1, "kind": "port",
           "sample": f"first {len(xyz)} source points, k={k}, resolution {res} m, {cpu_s:.1f} s"}
    for name, fn in (("this_repo_parallel", engine.supervoxel_parallel), ("this_repo_identical", engine.supervoxel)):
        fn(dev_xyz, k, res)  # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        labels, K = fn(dev_xyz, k, res)
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t
        out[name] = {"value": round(len(xyz) / gpu_s / 1e6, 4), "unit": "Mpts/s", "seconds": round(gpu_s, 4), "n_supervoxels": int(K),
                     "labels_identical_to_the_port": bool(np.array_equal(labels.cpu().numpy(), ref["labels"])) and K == ref["n_supervoxels"],
                     "note": ("everything on the device; same count and invariants as the reference's algorithm, not its labels" if "parallel" in name
                              else "everything on the device: the reference's sequential fusion and FIFO exchange as fixed points of parallel "
                                   "passes (csrc/supervoxel_exact.hip); rounds 1-4 replayed them on one host core")}
    # the same at the size of one of the reference's tiles (<= 1 M points, fusion_brienz.yaml:25-26): the sample above is small
    # enough for the one-core port, and too small to fill the device
    big = src[:1_000_000]
    if big.shape[0] >= 1_000_000:
        out["this_repo_identical_1M"] = {}
        for name, fn in (("identical", engine.supervoxel), ("parallel", engine.supervoxel_parallel)):
            fn(big, k, res)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                labels, K = fn(big, k, res)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            out["this_repo_identical_1M"][name] = {"value": round(1.0 / min(ts), 3), "unit": "Mpts/s", "ms": round(1e3 * min(ts), 2), "n_supervoxels": int(K)}
    return out


def cpu_baseline(prob, budget_s):
    """The oracle (single-thread C port of the same step) on a bounded prefix of the rank's patches."""
    import numpy as np

    from oracle import oracle as O

    P = prob.d["P"]
    pts_per_patch = prob.d["n_src"] / max(P, 1)

    def run(h, p_hi):
        t = time.perf_counter()
        R, tt = O.kabsch_batched(h["cs"], h["ct"], h["coff"], eps=1e-6)
        T0 = np.tile(np.eye(4), (p_hi, 1, 1))
        T0[:, :3, :3] = R
        T0[:, :3, 3] = tt
        res = O.piecewise_icp(h["src"], h["src_off"], h["tgt"], h["tgt_off"], init_T=T0, max_corr_dist=MAX_CORR,
                              max_iter=MAX_ITER, fixed_iters=True)
        s = h["src"].astype(np.float64)
        pid = np.repeat(np.arange(p_hi), np.diff(h["src_off"]))
        _ = np.einsum("nij,nj->ni", res["T"][pid, :3, :3], s) + res["T"][pid, :3, 3]
        return time.perf_counter() - t, int(h["src"].shape[0])

    probe = min(P, 16)
    dt, npts = run(prob.host_prefix(probe), probe)
    for _ in range(2):  # calibrate twice: the first patches of a cloud are border patches and not typical
        rate = npts / dt
        probe = int(min(P, max(probe, 0.05 * budget_s * rate / pts_per_patch)))
        dt, npts = run(prob.host_prefix(probe), probe)
    rate = npts / dt
    want = int(min(P, max(probe, 0.7 * budget_s * rate / pts_per_patch)))
    dt, npts = run(prob.host_prefix(want), want)
    out = {"value": round(npts / dt / 1e6, 5), "unit": "Mpts/s", "cores": 1, "kind": "port",
           "sample": f"first {want} of {P} patches ({npts} source points), same Kabsch+ICP(20)+apply step, {dt:.1f} s"}
    # the same port with its patch loop on every core this process may use (OpenMP)
    cores = host_cores()
    if cores > 1:
        used = O.set_threads(cores)
        try:
            want_all = int(min(P, max(probe, want * used * 0.5)))
            h = prob.host_prefix(want_all)
            run(prob.host_prefix(min(want_all, 4 * used)), min(want_all, 4 * used))  # thread start-up outside the clock
            dt_all, npts_all, reps = 0.0, 0, 0
            while dt_all < 0.3 * budget_s and reps < 20:
                dt1, n1 = run(h, want_all)
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
