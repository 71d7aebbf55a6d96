"""Multi-GPU: patches are independent units (every iteration of the reference's per-patch loop,
src/coarse_to_fine_matching_base.py:3254, touches only its own two index sets), so they shard across the ranks of
one node with NO data-path collective; the only exchange is one all-gather of the per-patch results
(4x4 transform + fitness + rmse + iterations = 19 doubles = 152 B per patch) over RCCL/xGMI.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  The reference has
no multi-GPU code at all: this module is new design, covered by world_size-2 gloo tests.
"""
import numpy as np


def lpt_assign(sizes, world):
    """Longest-processing-time-first: patches sorted by cost, each to the currently lightest rank.
    sizes: (P,) cost per patch (points x points for brute-force search).  Returns list of int64 arrays (patch ids per
    rank, ascending)."""
    sizes = np.asarray(sizes, dtype=np.float64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world)
    owner = np.empty(len(sizes), dtype=np.int64)
    for p in order:
        r = int(np.argmin(load))
        owner[p] = r
        load[r] += sizes[p]
    return [np.nonzero(owner == r)[0] for r in range(world)]


def take_patches(pts, off, ids):
    """CSR sub-selection on host arrays: points of the patches `ids`, re-packed contiguously."""
    off = np.asarray(off)
    cnt = off[ids + 1] - off[ids]
    new_off = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(cnt, out=new_off[1:])
    if len(ids):
        idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in ids]) if cnt.sum() else np.zeros(0, np.int64)
    else:
        idx = np.zeros(0, np.int64)
    return np.ascontiguousarray(pts[idx]), new_off


def gather_patch_results(local, ids_per_rank, rank, world, P, device):
    """All-gather of per-patch results.  local: dict(T (p,4,4), fitness (p,), rmse (p,), iters (p,)) torch tensors of
    this rank, rows in the order of ids_per_rank[rank].  Returns the same dict for ALL P patches in global order on
    every rank.  Ranks pad to the largest share so that a plain (fixed-size) all_gather can be used."""
    import torch
    import torch.distributed as dist
    pmax = max(len(i) for i in ids_per_rank)
    packed = torch.zeros((pmax, 19), dtype=torch.float64, device=device)
    p = len(ids_per_rank[rank])
    if p:
        packed[:p, :16] = local["T"].reshape(p, 16).to(torch.float64)
        packed[:p, 16] = local["fitness"].to(torch.float64)
        packed[:p, 17] = local["rmse"].to(torch.float64)
        packed[:p, 18] = local["iters"].to(torch.float64)
    if world > 1:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed)
    else:
        parts = [packed]
    full = torch.zeros((P, 19), dtype=torch.float64, device=device)
    for r in range(world):
        ids = torch.from_numpy(ids_per_rank[r]).to(device)
        full[ids] = parts[r][:len(ids_per_rank[r])]
    return dict(T=full[:, :16].reshape(P, 4, 4), fitness=full[:, 16], rmse=full[:, 17], iters=full[:, 18].to(torch.int32))


def piecewise_icp_sharded(src, src_off, tgt, tgt_off, init_T=None, rank=0, world=1, device=None, compute_fn=None, **icp_kw):
    """Shard P patches over `world` ranks (LPT on ns x nt), run the per-patch ICP of this rank's share, all-gather.

    src/tgt/offsets/init_T are HOST numpy arrays (identical on every rank, e.g. loaded from the same tile files);
    each rank uploads only its own patches.  compute_fn(src, src_off, tgt, tgt_off, init_T, **icp_kw) -> dict of
    torch tensors is the per-rank engine; the default is the HIP path (engine.piecewise_icp) -- the CPU tests inject
    a checker instead."""
    import torch
    src_off, tgt_off = np.asarray(src_off), np.asarray(tgt_off)
    P = len(src_off) - 1
    cost = np.diff(src_off).astype(np.float64) * np.maximum(np.diff(tgt_off), 1)
    ids_per_rank = lpt_assign(cost, world)
    mine = ids_per_rank[rank]
    s, so = take_patches(src, src_off, mine)
    t, to = take_patches(tgt, tgt_off, mine)
    T0 = None if init_T is None else np.ascontiguousarray(np.asarray(init_T)[mine])
    if compute_fn is None:
        from . import engine
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())

        def compute_fn(s, so, t, to, T0, **kw):
            up = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
            return engine.piecewise_icp(up(s), up(so), up(t), up(to), init_T=None if T0 is None else up(T0),
                                        max_src_patch=int(np.diff(so).max()) if len(so) > 1 else 0,
                                        max_tgt_patch=int(np.diff(to).max()) if len(to) > 1 else 0, **kw)
    if device is None:
        device = torch.device("cpu")
    if len(mine):
        local = compute_fn(s, so, t, to, T0, **icp_kw)
    else:
        local = dict(T=torch.zeros((0, 4, 4), dtype=torch.float64), fitness=torch.zeros(0), rmse=torch.zeros(0),
                     iters=torch.zeros(0, dtype=torch.int32))
    return gather_patch_results(local, ids_per_rank, rank, world, P, device), ids_per_rank


class TileResultGather:
    """Weak-scaling exchange of `bench.py --gpus N`: every rank owns whole tiles and all-gathers the per-patch results
    (4x4 transform, fitness, rmse, iterations = 19 doubles per patch) of the tile it just finished.  Two sets of receive
    buffers, so that the collective of step i runs while step i + 1 computes; `submit` blocks only when the set it is
    about to reuse is still in flight, `drain` waits for everything.  Backend-agnostic (RCCL on the GPUs, gloo in the
    CPU test)."""

    def __init__(self, dist, torch, world, n_patches, device):
        self.dist, self.torch, self.world, self.P = dist, torch, world, n_patches
        self.sets = [[torch.empty((n_patches, 19), dtype=torch.float64, device=device) for _ in range(world)] for _ in range(2)]
        self.inflight = [None, None]  # (work handle, packed tensor kept alive) per buffer set
        self.count = 0

    def pack(self, out):
        torch = self.torch
        return torch.cat([out["T"].reshape(self.P, 16), out["fitness"][:, None].to(torch.float64),
                          out["rmse"][:, None].to(torch.float64), out["iters"].to(torch.float64)[:, None]], dim=1)

    def submit(self, out):
        slot = self.count % 2
        self.count += 1
        if self.inflight[slot] is not None:
            self.inflight[slot][0].wait()  # the buffer set is free again
        packed = self.pack(out)
        self.inflight[slot] = (self.dist.all_gather(self.sets[slot], packed, async_op=True), packed)
        return slot

    def drain(self):
        for w in self.inflight:
            if w is not None:
                w[0].wait()

    def latest(self):
        """The gathered results of the most recent submit (list over ranks of (P, 19) tensors); call after drain()."""
        return self.sets[(self.count - 1) % 2]
