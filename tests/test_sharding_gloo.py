"""CPU suite, part 3: the N > 1 path (patch sharding + all-gather of per-patch results) with world_size 2 on gloo.
The per-rank engine is injected (the oracle acts as the checker here); on a GPU box the default engine is the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_engine(s, so, t, to, T0, **kw):
    from oracle import oracle as O
    r = O.piecewise_icp(s, so, t, to, init_T=T0, **kw)
    return dict(T=torch.from_numpy(r["T"]), fitness=torch.from_numpy(r["fitness"]), rmse=torch.from_numpy(r["rmse"]),
                iters=torch.from_numpy(r["iters"]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd import sharding, synthetic
    d = synthetic.make_patches(6000, 4, 1.386, seed=2)
    out, ids = sharding.piecewise_icp_sharded(d["src"], d["src_off"], d["tgt"], d["tgt_off"], rank=rank, world=world,
                                              compute_fn=_oracle_engine, max_corr_dist=0.1, max_iter=20, fixed_iters=True)
    q.put((rank, out["T"].numpy(), out["fitness"].numpy(), out["iters"].numpy(), [i.tolist() for i in ids]))
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_assign_balances_and_covers():
    from fusion4landslide_amd import sharding
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 1000, 200)
    for world in (1, 2, 3, 8):
        parts = sharding.lpt_assign(sizes, world)
        allp = np.sort(np.concatenate(parts))
        assert np.array_equal(allp, np.arange(200))
        loads = np.array([sizes[p].sum() for p in parts])
        assert loads.max() - loads.min() <= sizes.max()
    pts = rng.normal(size=(10, 3))
    off = np.array([0, 3, 3, 7, 10])
    sub, so = sharding.take_patches(pts, off, np.array([2, 0]))
    assert np.array_equal(so, [0, 4, 7]) and np.array_equal(sub, np.concatenate([pts[3:7], pts[0:3]]))
    sub_t, so_t = sharding.take_patches(torch.from_numpy(pts), torch.from_numpy(off), np.array([2, 0]))
    assert np.array_equal(so_t.numpy(), so) and np.array_equal(sub_t.numpy(), sub)
    sub, so = sharding.take_patches(pts, off, np.array([1]))
    assert sub.shape == (0, 3) and np.array_equal(so, [0, 0])


def test_lpt_assign_and_take_patches_properties():
    """Random patch costs (zeros, ties, one giant among dwarfs, fewer patches than ranks) and worlds 1..9: every patch goes to
    exactly one rank, ids ascending per rank, the greedy bound max - min <= largest cost, the same answer twice; and
    take_patches re-packs exactly the chosen patches' rows, on numpy arrays and torch tensors alike."""
    from hypothesis import given, settings, strategies as st
    from fusion4landslide_amd import sharding

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.integers(0, 10**6), min_size=0, max_size=120), st.integers(1, 9), st.integers(0, 2**31 - 1))
    def prop(costs, world, seed):
        sizes = np.asarray(costs, dtype=np.int64)
        parts = sharding.lpt_assign(sizes, world)
        assert len(parts) == world and all(np.all(np.diff(p) > 0) for p in parts)
        allp = np.sort(np.concatenate(parts)) if len(sizes) else np.zeros(0, np.int64)
        assert np.array_equal(allp, np.arange(len(sizes)))
        loads = np.array([sizes[p].sum() for p in parts])
        assert loads.max() - loads.min() <= (sizes.max() if len(sizes) else 0)
        again = sharding.lpt_assign(sizes, world)
        assert all(np.array_equal(a, b) for a, b in zip(parts, again))
        # the rows of a rank's patches, re-packed
        rng = np.random.default_rng(seed)
        cnt = rng.integers(0, 5, len(sizes))
        off = np.zeros(len(sizes) + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        pts = rng.normal(size=(int(off[-1]), 3))
        for ids in parts[:2]:
            sub, so = sharding.take_patches(pts, off, ids)
            want = [pts[off[i]:off[i + 1]] for i in ids]
            assert np.array_equal(np.diff(so), cnt[ids]) and np.array_equal(sub, np.concatenate(want) if want else np.zeros((0, 3)))
            sub_t, so_t = sharding.take_patches(torch.from_numpy(pts), torch.from_numpy(off), ids)
            assert np.array_equal(sub_t.numpy(), sub) and np.array_equal(so_t.numpy(), so)

    prop()


@pytest.mark.timeout(300)
def test_sharded_icp_world2_matches_single_process():
    from fusion4landslide_amd import synthetic
    from oracle import oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = synthetic.make_patches(6000, 4, 1.386, seed=2)
    ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=20, fixed_iters=True)
    res.sort(key=lambda r: r[0])
    for rank, T, fit, iters, ids in res:
        assert np.array_equal(T, ref["T"]) and np.array_equal(fit, ref["fitness"]) and (iters == 20).all()
    ids0 = res[0][4]
    assert sorted(ids0[0] + ids0[1]) == list(range(16)) and ids0 == res[1][4]


def _oracle_patch_loop(d):
    """Checker standing in for the per-rank engine (f4l_patch_loop) on torch CPU tensors."""
    from oracle import oracle as O
    r = O.piecewise_icp(d["src"].numpy(), d["src_off"].numpy(), d["tgt"].numpy(), d["tgt_off"].numpy(), max_corr_dist=0.1,
                        max_iter=20, fixed_iters=True)
    return dict(T=torch.from_numpy(r["T"]), fitness=torch.from_numpy(r["fitness"]), rmse=torch.from_numpy(r["rmse"]),
                iters=torch.from_numpy(r["iters"]))


def _gather_worker(rank, world, port, q):
    """The code path of `bench.py --gpus N`: the same cloud built by every rank, shard_cloud, a step per rank, the
    double-buffered all-gather, results read back in global patch order."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd import sharding, synthetic
    cloud = synthetic.make_patches_device(8000, 5, 1.386, torch.device("cpu"), seed=4)
    d, ids = sharding.shard_cloud(cloud, rank, world)
    g = sharding.PatchResultGather(dist, torch, ids, rank, torch.device("cpu"))
    out = _oracle_patch_loop(d)
    seen = []
    for step in range(5):  # more steps than buffer sets: the third submit must wait for the first
        o = dict(out, rmse=out["rmse"] + step)
        g.submit(o)
        if step % 2 == 1:
            g.drain()
            seen.append({k: v.clone().numpy() for k, v in g.latest().items()})
    g.drain()
    seen.append({k: v.clone().numpy() for k, v in g.latest().items()})
    q.put((rank, seen, [i.tolist() for i in ids], d["P"], d["n_src"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_cloud_world2_gathers_global_order():
    """Two gloo ranks shard ONE cloud, run their shares and all-gather: every rank ends up with every patch's result in
    global patch order, equal to the single-process run, for every step looked at."""
    from fusion4landslide_amd import sharding, synthetic
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cloud = synthetic.make_patches_device(8000, 5, 1.386, torch.device("cpu"), seed=4)
    ref = _oracle_patch_loop(cloud)
    assert sorted(res[0][2][0] + res[0][2][1]) == list(range(25)) and res[0][2] == res[1][2]
    assert res[0][3] + res[1][3] == 25 and res[0][4] + res[1][4] == 8000
    costs = sharding.patch_costs(cloud["src_off"], cloud["tgt_off"])
    loads = [costs[np.array(i)].sum() for i in res[0][2]]
    assert abs(loads[0] - loads[1]) <= costs.max()
    for rank, seen, ids, P, n in res:
        for k, step in enumerate((1, 3, 4)):
            assert np.array_equal(seen[k]["T"], ref["T"].numpy()) and np.array_equal(seen[k]["fitness"], ref["fitness"].numpy())
            assert np.array_equal(seen[k]["rmse"], ref["rmse"].numpy() + step) and (seen[k]["iters"] == 20).all()


@pytest.mark.timeout(300)
def test_bench_gpus2_dry_run_starts_two_ranks():
    """`python bench.py --gpus 2` without a launcher around it starts two ranks itself (orchestration dry run: gloo, CPU
    tensors, a stub launch) and prints ONE line with n_gpus == 2; a WORLD_SIZE that disagrees with --gpus is an error."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--config", "C1_50k_64",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] and line["dry_run_gather_ok"] and line["value"] is None
    assert line["scaling"] == "strong" and line["config"]["patches"] == 64 and line["config"]["patches_on_rank0"] == 32
    assert line["config"]["dist_backend"] == "gloo" and line["config"]["dist_world_size"] == 2  # (nccl = RCCL on the GPUs)
    # every rank's own numbers, not only the slowest rank's clock: LPT balance and the all-gather by itself
    pr = line["per_rank"]
    assert pr["patches"] == {"min": 32.0, "max": 32.0, "mean": 32.0} and pr["points"]["min"] > 0
    assert abs(pr["points"]["mean"] * 2 - line["config"]["points_per_epoch"]) < 1
    assert pr["points"]["max"] <= 1.1 * pr["points"]["mean"]                      # LPT: no rank far above its share
    assert pr["allgather_ms"]["min"] > 0 and pr["allgather_bytes_per_rank"] == 32 * 19 * 8
    assert pr["wall_ms_per_step"]["max"] <= line["ms_per_step"] * 1.0001 + 1e-3   # the line's time is the slowest rank's
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4"), timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_rank_share_generated_alone_equals_the_share_of_the_whole_cloud():
    """synthetic.make_rank_share_device: a rank's share of the bench cloud produced WITHOUT holding the whole cloud (counting pass,
    LPT on the counts, keeping pass) is bit for bit `shard_cloud(make_patches_device(...), rank, world)` -- points, offsets, patch
    ids --, for chunk sizes that do not divide the cloud."""
    import torch
    from fusion4landslide_amd import sharding, synthetic
    dev = torch.device("cpu")
    whole = synthetic.make_patches_device(120_000, 11, 1.386, dev, seed=3, chunk=50_000)
    for world in (1, 3, 8):
        seen = np.zeros(whole["P"], dtype=np.int64)
        for rank in range(world):
            a, ids_a = sharding.shard_cloud(whole, rank, world)
            b, ids_b = synthetic.make_rank_share_device(120_000, 11, 1.386, dev, rank, world, seed=3, chunk=47_000)
            assert all(np.array_equal(x, y) for x, y in zip(ids_a, ids_b))
            for k in ("src", "src_off", "tgt", "tgt_off"):
                assert torch.equal(a[k], b[k]), (world, rank, k)
            assert (a["P"], a["max_src"], a["max_tgt"], a["n_src"]) == (b["P"], b["max_src"], b["max_tgt"], b["n_src"])
            seen[ids_b[rank]] += 1
        assert (seen == 1).all()


def _share_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd import synthetic
    b, ids = synthetic.make_rank_share_device(120_000, 11, 1.386, torch.device("cpu"), rank, world, seed=3, chunk=17_000, dist=dist)
    q.put((rank, {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in b.items()}, [i.tolist() for i in ids]))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_share_with_the_counting_pass_shared_between_the_ranks():
    """make_rank_share_device(dist=...): every rank counts every world-th chunk of the generator and ONE all_reduce(SUM) of the
    per-patch counts gives all of them the whole cloud's (one counting pass per node instead of one per
    rank).  Three gloo ranks, eight chunks: every rank's share is still bit for bit the share of the whole cloud."""
    from fusion4landslide_amd import sharding, synthetic
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = synthetic.make_patches_device(120_000, 11, 1.386, torch.device("cpu"), seed=3, chunk=50_000)
    for rank, b, ids in got:
        a, ids_a = sharding.shard_cloud(whole, rank, world)
        assert all(np.array_equal(x, np.asarray(y)) for x, y in zip(ids_a, ids))
        for k in ("src", "src_off", "tgt", "tgt_off"):
            assert np.array_equal(a[k].numpy(), b[k]), (rank, k)


@pytest.mark.timeout(600)
def test_bench_gpus8_dry_run_is_one_line_of_eight_ranks():
    """The orchestration of an 8-GPU node (`--gpus 8 --dry-run`: eight gloo ranks on CPU tensors, every rank generating only its own
    share, the per-patch results of all of them arriving in global patch order on rank 0)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--config", "C1_50k_64",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["dry_run"] and line["dry_run_gather_ok"] and line["value"] is None
    assert line["config"]["patches"] == 64 and line["config"]["patches_on_rank0"] == 8
    assert line["config"]["dist_backend"] == "gloo" and line["config"]["dist_world_size"] == 8
