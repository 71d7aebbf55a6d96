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


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd.sharding import TileResultGather
    P = 7
    g = TileResultGather(dist, torch, world, P, torch.device("cpu"))
    seen = []
    for step in range(5):  # more steps than buffer sets: the third submit must wait for the first
        T = torch.eye(4, dtype=torch.float64).repeat(P, 1, 1) * (100 * step + 10 * rank + 1)
        out = dict(T=T, fitness=torch.full((P,), step + 0.5 * rank, dtype=torch.float64),
                   rmse=torch.full((P,), 0.25 * step, dtype=torch.float64), iters=torch.full((P,), 20 + rank, dtype=torch.int32))
        g.submit(out)
        if step % 2 == 1:  # look at a finished step now and then
            g.drain()
            seen.append([t.clone() for t in g.latest()])
    g.drain()
    seen.append([t.clone() for t in g.latest()])
    q.put((rank, [[t.numpy() for t in s] for s in seen]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tile_result_gather_world2_double_buffered():
    """The exchange of `bench.py --gpus N` (every rank all-gathers the per-patch results of its tile, two buffer sets in
    flight) with two gloo ranks: every rank ends up with every rank's rows, for the steps looked at."""
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
    for rank, seen in res:
        for k, step in enumerate((1, 3, 4)):
            for r in range(world):
                row = seen[k][r]
                assert row.shape == (7, 19)
                assert row[0, 0] == 100 * step + 10 * r + 1 and row[0, 5] == 100 * step + 10 * r + 1 and row[0, 1] == 0
                assert (row[:, 16] == step + 0.5 * r).all() and (row[:, 17] == 0.25 * step).all() and (row[:, 18] == 20 + r).all()
