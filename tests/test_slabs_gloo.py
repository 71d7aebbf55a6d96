"""CPU suite: the multi-GPU split of the partition stage (fusion4landslide_amd/slabs.py: slabs along x, halo exchange, seam
ownership) with world_size 2 and 3 on gloo.  The per-rank kernels are injected (the oracle's kNN / normals and the numpy model
of the device segmentation act as the checkers here); on a GPU node the defaults are the HIP path over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
N, K_NN, RES, HALO = 9000, 12, 0.5, 0.4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cloud(n=N, lx=6.0):
    rng = np.random.default_rng(11)
    xy = rng.uniform(0, 1, (n, 2)) * [lx, 3.0]
    z = 0.2 * np.sin(2 * xy[:, 0]) * np.cos(3 * xy[:, 1]) + rng.normal(0, 0.003, n)
    return np.c_[xy, z].astype(np.float32)


def _knn_normals(p):
    from oracle import oracle as O
    idx, d2 = O.knn(p.numpy(), K_NN)
    return torch.from_numpy(idx), torch.from_numpy(d2), torch.from_numpy(O.normals_from_knn(p.numpy(), idx))


def _segment(p, normals, knn, res, grid_bbox):
    from oracle import sv_parallel as M
    r = M.segment(p.numpy(), normals.numpy(), knn.numpy(), res, grid_bbox=grid_bbox)
    assert r["status"] == 0
    return torch.from_numpy(r["labels"]), r["n_supervoxels"]


def _target_cloud(n=N, lx=6.0):
    """The second epoch: the same surface sampled elsewhere and displaced by a few centimetres (more than the point spacing)."""
    rng = np.random.default_rng(12)
    xy = rng.uniform(0, 1, (n + 500, 2)) * [lx, 3.0]
    z = 0.2 * np.sin(2 * xy[:, 0]) * np.cos(3 * xy[:, 1]) + rng.normal(0, 0.003, n + 500)
    return (np.c_[xy, z] + [0.07, -0.05, 0.02]).astype(np.float32)


def _nn(cloud, queries):
    from scipy.spatial import cKDTree
    d, i = cKDTree(cloud.numpy().astype(np.float64)).query(queries.numpy().astype(np.float64), k=1)
    return torch.from_numpy(i.astype(np.int64)), torch.from_numpy(d * d)


def _worker(rank, world, port, q, n=N, lx=6.0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd import slabs
    xyz = _cloud(n, lx)
    mine = np.arange(rank, n, world)  # an arbitrary chunk per rank: every world-th point
    out = slabs.slab_supervoxel(torch.from_numpy(xyz[mine]), torch.from_numpy(mine.astype(np.int64)), K_NN, RES, dist, rank, world, HALO,
                                knn_normals_fn=_knn_normals, segment_fn=_segment)
    tgt = _target_cloud(n, lx)
    tmine = np.arange(world - 1 - rank, len(tgt), world)
    tg = slabs.slab_targets(torch.from_numpy(tgt[tmine]), out, dist, rank, world, HALO, nn_fn=_nn)
    out["tgt_xyz"], out["tgt_src_gid"], out["tgt_d2"] = tg["xyz"], out["gid"][tg["nn"]], tg["d2"]
    out["tgt_forwarded"], out["tgt_uncertified"] = tg["n_forwarded"], tg["n_uncertified"]
    q.put((rank, {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in out.items() if k != "plan"}, out["plan"]["bounds"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_split_matches_the_whole_cloud(world):
    """(world 8: the split of a whole node, on a cloud long enough along x for eight slabs wider than the halo.)"""
    from oracle import oracle as O
    from oracle import sv_parallel as M
    N, lx = (9000, 6.0) if world < 8 else (24000, 16.0)  # noqa: N806 (shadows the module's default on purpose)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, N, lx)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz = _cloud(N, lx)
    gidx, gd2 = O.knn(xyz, K_NN)
    bounds = res[0][2]
    assert all(r[2] == bounds for r in res) and bounds[0] == 0 and sorted(bounds) == bounds
    # ownership: every point on exactly one rank, on the rank whose columns hold it; balanced
    gids = np.concatenate([r[1]["gid"] for r in res])
    assert np.array_equal(np.sort(gids), np.arange(N))
    x0 = xyz[:, 0].min()
    for rank, out, _ in res:
        col = np.clip(((out["xyz"][:, 0].astype(np.float64) - x0) / RES).astype(int), 0, bounds[-1] - 1)
        assert ((col >= bounds[rank]) & (col < bounds[rank + 1])).all()
        assert np.array_equal(out["xyz"], xyz[out["gid"]])
        assert abs(len(out["gid"]) - N / world) < 0.25 * N / world
        assert out["n_uncertified"] == 0 and (out["n_halo"] > 0) == (world > 1)
        # the owned points' neighbour lists ARE the whole cloud's (ids and squared distances)
        assert np.array_equal(out["d2"], gd2[out["gid"]])
        assert np.array_equal(out["knn_gid"], gidx[out["gid"]])
    # supervoxels: the slabs' counts add up to the whole cloud's occupied cells; labels contiguous; none crosses a cut
    K = M.occupied_cells(xyz, RES)
    assert all(r[1]["K_total"] == K for r in res) and sum(r[1]["K_local"] for r in res) == K
    labels = np.empty(N, dtype=np.int64)
    for rank, out, _ in res:
        labels[out["gid"]] = out["labels"]
        assert out["labels"].min() == out["offset"] and out["labels"].max() == out["offset"] + out["K_local"] - 1
    assert np.array_equal(np.unique(labels), np.arange(K))
    owner = np.empty(N, dtype=np.int64)
    for rank, out, _ in res:
        owner[out["gid"]] = rank
    for lab in range(K):
        assert len(set(owner[labels == lab])) == 1
    # the second epoch: every target point ends on exactly one rank, the one that owns its nearest source point of the WHOLE
    # first epoch (so that it joins a patch that lives there); points near a cut were forwarded
    from scipy.spatial import cKDTree
    tgt = _target_cloud(N, lx)
    d_ref, i_ref = cKDTree(xyz.astype(np.float64)).query(tgt.astype(np.float64), k=1)
    got_xyz = np.concatenate([r[1]["tgt_xyz"] for r in res])
    got_gid = np.concatenate([r[1]["tgt_src_gid"] for r in res])
    got_d2 = np.concatenate([r[1]["tgt_d2"] for r in res])
    assert got_xyz.shape == tgt.shape
    o_got, o_ref = np.lexsort(got_xyz.T[::-1]), np.lexsort(tgt.T[::-1])
    assert np.array_equal(got_xyz[o_got], tgt[o_ref])
    assert np.array_equal(got_gid[o_got], i_ref[o_ref])
    assert np.abs(np.sqrt(got_d2[o_got]) - d_ref[o_ref]).max() < 1e-12
    assert all(r[1]["tgt_uncertified"] == 0 for r in res) and sum(r[1]["tgt_forwarded"] for r in res) > 0
    for rank, out, _ in res:
        assert (owner[out["tgt_src_gid"]] == rank).all()


def _worker_heavy_column(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fusion4landslide_amd import slabs
    rng = np.random.default_rng(3)
    n = 6000
    x = np.where(rng.uniform(size=n) < 0.7, rng.uniform(2.0, 2.0 + RES * 0.9, n), rng.uniform(0, 5, n))  # 70 % of the points in ONE grid column
    xyz = np.c_[x, rng.uniform(0, 3, n), rng.normal(0, 0.01, n)].astype(np.float32)
    mine = np.arange(rank, n, world)
    try:
        slabs.slab_supervoxel(torch.from_numpy(xyz[mine]), torch.from_numpy(mine.astype(np.int64)), K_NN, RES, dist, rank, world, 0.2,
                              knn_normals_fn=_knn_normals, segment_fn=_segment)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, str(e)))
    dist.barrier()  # (every rank raised at the same point: nobody is left inside a collective)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_slab_without_columns_is_refused_on_every_rank():
    """with 70 % of the points in one column of the resolution grid and three ranks, the balanced cuts leave one rank no
    column.  Halos only travel between adjacent ranks, so its neighbours would certify neighbour lists that miss the points on the
    other side: `slab_supervoxel` must refuse, on every rank together (decided from the bounds all ranks hold)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_heavy_column, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == world and all("own no grid column" in msg for msg in res.values()), res


def test_no_bitwise_reduction_reaches_a_backend_that_lacks_it():
    """RCCL has no ReduceOp.BOR / BAND / BXOR (torch: "Cannot use ReduceOp.BOR with NCCL"); the slab path ran only over gloo
    until round 4 and carried one.  `_all_reduce` refuses them wherever the backend is not gloo, and the flag
    exchange of `_raise_together` is a MAX over the flags' bits: the OR, on any backend."""
    from fusion4landslide_amd import slabs

    class FakeDist:
        ReduceOp = dist.ReduceOp

        def __init__(self, backend, others):
            self.backend, self.others, self.ops = backend, others, []

        def get_backend(self):
            return self.backend

        def all_reduce(self, t, op=None):
            self.ops.append(op)
            assert op == dist.ReduceOp.MAX
            for o in self.others:  # what the other ranks contribute
                t.copy_(torch.maximum(t, o))

    for op in (dist.ReduceOp.BOR, dist.ReduceOp.BAND, dist.ReduceOp.BXOR):
        with pytest.raises(ValueError):
            slabs._all_reduce(FakeDist("nccl", []), torch.zeros(1, dtype=torch.int64), op)
    bits = lambda v: torch.tensor([(v >> b) & 1 for b in range(slabs._FLAG_BITS)], dtype=torch.int64)  # noqa: E731
    quiet = FakeDist("nccl", [bits(0), bits(0)])
    slabs._raise_together(quiet, torch, torch.device("cpu"), 3, 0, "nothing")
    assert quiet.ops == [dist.ReduceOp.MAX]
    loud = FakeDist("nccl", [bits(4), bits(2)])
    with pytest.raises(RuntimeError, match=r"flags over all ranks: 7\]"):
        slabs._raise_together(loud, torch, torch.device("cpu"), 3, 1, "three ranks, three different bits")
