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
    import heapq
    sizes = np.asarray(sizes, dtype=np.float64)
    order = np.argsort(-sizes, kind="stable")
    owner = np.empty(len(sizes), dtype=np.int64)
    heap = [(0.0, r) for r in range(world)]  # (load, rank): ties go to the lowest rank, like argmin
    for p, c in zip(order.tolist(), sizes[order].tolist()):
        load, r = heap[0]
        owner[p] = r
        heapq.heapreplace(heap, (load + c, r))
    return [np.nonzero(owner == r)[0] for r in range(world)]


def take_patches(pts, off, ids):
    """CSR sub-selection: the points of the patches `ids`, re-packed contiguously, and their new offsets.  Works on host
    numpy arrays and on torch tensors of any device alike (no per-patch loop: one repeat + one gather)."""
    if isinstance(pts, np.ndarray):
        off, ids = np.asarray(off), np.asarray(ids, dtype=np.int64)
        cnt = off[ids + 1] - off[ids]
        new_off = np.zeros(len(ids) + 1, dtype=np.int64)
        np.cumsum(cnt, out=new_off[1:])
        idx = np.repeat(off[ids] - new_off[:-1], cnt) + np.arange(new_off[-1], dtype=np.int64)
        return np.ascontiguousarray(pts[idx]), new_off
    import torch
    ids = torch.as_tensor(ids, dtype=torch.int64, device=pts.device)
    cnt = off[ids + 1] - off[ids]
    new_off = torch.zeros(ids.shape[0] + 1, dtype=torch.int64, device=pts.device)
    new_off[1:] = torch.cumsum(cnt, 0)
    total = int(new_off[-1].item())
    idx = torch.repeat_interleave(off[ids] - new_off[:-1], cnt, output_size=total) + torch.arange(total, dtype=torch.int64, device=pts.device)
    return pts[idx].contiguous(), new_off


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


def patch_costs(src_off, tgt_off):
    """Cost model of one patch for the assignment: source points x target points (what a search without any pruning
    would touch); host float64 array from host or device offsets."""
    so = src_off.cpu().numpy() if hasattr(src_off, "cpu") else np.asarray(src_off)
    to = tgt_off.cpu().numpy() if hasattr(tgt_off, "cpu") else np.asarray(tgt_off)
    return np.diff(so).astype(np.float64) * np.maximum(np.diff(to), 1)


def shard_cloud(d, rank, world):
    """This rank's share of ONE two-epoch cloud `d` (dict src, src_off, tgt, tgt_off; numpy or torch on any device; the
    same on every rank): LPT assignment of its patches, then the points of the rank's own patches re-packed.  Returns
    (share dict with the same keys + P, max_src, max_tgt, n_src; ids_per_rank)."""
    ids_per_rank = lpt_assign(patch_costs(d["src_off"], d["tgt_off"]), world)
    mine = ids_per_rank[rank]
    s, so = take_patches(d["src"], d["src_off"], mine)
    t, to = take_patches(d["tgt"], d["tgt_off"], mine)
    mx = lambda o: int((o[1:] - o[:-1]).max()) if len(mine) else 0  # noqa: E731
    return dict(src=s, src_off=so, tgt=t, tgt_off=to, P=len(mine), max_src=mx(so), max_tgt=mx(to), n_src=int(so[-1])), ids_per_rank


class PatchResultGather:
    """The exchange step of the sharded path (`bench.py --gpus N`): every rank all-gathers the per-patch
    results of its share (4x4 transform, fitness, rmse, iterations = 19 doubles = 152 B per patch; shares padded to the
    largest one so that a plain fixed-size all-gather serves) and reads them back in GLOBAL patch order through one
    precomputed gather.  Two sets of receive buffers: the collective of step i runs while step i + 1 computes; `submit`
    blocks only when the set it is about to reuse is still in flight, `drain` waits for everything.  Backend-agnostic
    (RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, dist, torch, ids_per_rank, rank, device):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, len(ids_per_rank)
        self.p = len(ids_per_rank[rank])
        self.pmax = max(1, max(len(i) for i in ids_per_rank))
        self.P = sum(len(i) for i in ids_per_rank)
        self.bufs = [torch.zeros((self.world * self.pmax, 19), dtype=torch.float64, device=device) for _ in range(2)]
        self.packed = [torch.zeros((self.pmax, 19), dtype=torch.float64, device=device) for _ in range(2)]
        pos = np.empty(self.P, dtype=np.int64)  # row of global patch g inside a receive buffer
        for r, ids in enumerate(ids_per_rank):
            pos[ids] = r * self.pmax + np.arange(len(ids))
        self.pos = torch.from_numpy(pos).to(device)
        self.inflight = [None, None]
        self.count = 0

    def submit(self, out):
        torch = self.torch
        slot = self.count % 2
        self.count += 1
        if self.inflight[slot] is not None:
            self.inflight[slot].wait()  # the buffer set is free again
        pk, p = self.packed[slot], self.p
        if p:
            pk[:p, :16] = out["T"].reshape(p, 16)
            pk[:p, 16], pk[:p, 17], pk[:p, 18] = out["fitness"], out["rmse"], out["iters"].to(torch.float64)
        if self.world > 1:
            # ONE flat receive buffer (rank r's rows at r * pmax): all_gather_into_tensor writes it in place -- an all_gather into a
            # LIST of views makes c10d gather into a staging buffer of its own and copy every view back
            self.inflight[slot] = self.dist.all_gather_into_tensor(self.bufs[slot], pk, async_op=True)
        else:
            self.bufs[slot][:self.pmax].copy_(pk)
        return slot

    def drain(self):
        for w in self.inflight:
            if w is not None:
                w.wait()
        self.inflight = [None, None]

    def latest(self):
        """Results of the most recent submit for ALL patches in global order; call after drain()."""
        torch = self.torch
        full = self.bufs[(self.count - 1) % 2][self.pos]
        return dict(T=full[:, :16].reshape(self.P, 4, 4), fitness=full[:, 16], rmse=full[:, 17], iters=full[:, 18].to(torch.int32))
