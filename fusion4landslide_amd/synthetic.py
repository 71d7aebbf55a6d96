"""Seeded synthetic two-epoch clouds for tests and bench.py.

Epoch 1: N points, (x, y) ~ U([0, L]^2), z = sum_j a_j sin(2 pi f_j x + phi_j) sin(2 pi g_j y + psi_j) with
a_j = 2 / 2^j m, f_j = g_j = 2^j / 100 1/m (j = 1..4), plus N(0, 5 mm) range noise.
Epoch 2: independently re-sampled on the same surface, then moved by a piecewise-rigid field: square blocks
of side 4 x resolution, each with a rotation <= 0.5 deg about a random axis through the block centre and a
translation U(-0.05, 0.05) m ("stable", 70 % of the blocks) or U(0.2, 0.5) m with random sign (the rest),
plus N(0, 5 mm).  float32 storage.

The ICP-timing partition is the (x, y) grid at `resolution` (independent of the supervoxel implementation):
patch c holds the source points of cell c and the target points of cell c, as CSR arrays.

Everything here is host numpy (PCG64) so that the CPU oracle and the GPU see bit-identical inputs; it is data
generation, not part of the measured path.
"""
import numpy as np

# configs of BASELINE.json: name -> (N per epoch, grid cells per side, resolution [m])
CONFIGS = {
    "C1_50k_64": dict(n=50_000, cells=8, resolution=1.386),
    "C2_1M_2k": dict(n=1_000_000, cells=45, resolution=1.386),
    "C2x4_4M_8k": dict(n=4_000_000, cells=90, resolution=1.386),     # C2 density, four times the area (scaling runs)
    "C2x16_16M_32k": dict(n=16_000_000, cells=180, resolution=1.386),
    "C3_10M_20k": dict(n=10_000_000, cells=141, resolution=0.1),
    "C4_50M_100k": dict(n=50_000_000, cells=316, resolution=1.386),
    # BASELINE.json configs[4]: the full hot path (supervoxel partition + per-patch loop) on 100 M points per epoch, C2's density;
    # `cells` only sizes the cloud and its block motion field here -- the patches are the supervoxels the path itself cuts
    "C5_100M_full": dict(n=100_000_000, cells=450, resolution=1.386),
}


def _surface(x, y, phases, roughness=0.0):
    z = np.zeros_like(x)
    for j in range(1, 5):
        a, f = 2.0 / 2 ** j, 2 ** j / 100.0
        z += a * np.sin(2 * np.pi * f * x + phases[j - 1, 0]) * np.sin(2 * np.pi * f * y + phases[j - 1, 1])
    if roughness:
        # metre-scale relief so that a single patch constrains all six degrees of freedom (tests only)
        z += roughness * np.sin(2 * np.pi * x / 0.9 + phases[0, 0]) * np.sin(2 * np.pi * y / 1.1 + phases[0, 1])
    return z


def _rodrigues(axis, angle):
    axis = axis / np.linalg.norm(axis, axis=1, keepdims=True)
    K = np.zeros((axis.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -axis[:, 2], axis[:, 1]
    K[:, 1, 0], K[:, 1, 2] = axis[:, 2], -axis[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -axis[:, 1], axis[:, 0]
    s, c = np.sin(angle)[:, None, None], np.cos(angle)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def two_epoch_cloud(n, cells, resolution, noise=0.005, seed=0, origin=(0.0, 0.0, 0.0), roughness=0.0):
    """Returns dict(src (n,3) f32, tgt (n,3) f32, L, block_R, block_t, block_of_tgt)."""
    L = cells * resolution
    phases = np.random.Generator(np.random.PCG64(seed)).uniform(0, 2 * np.pi, (4, 2))
    r0 = np.random.Generator(np.random.PCG64(seed))
    r0.uniform(0, 2 * np.pi, (4, 2))  # keep the stream aligned with `phases`
    xy = r0.uniform(0, L, (n, 2))
    src = np.c_[xy, _surface(xy[:, 0], xy[:, 1], phases, roughness) + r0.normal(0, noise, n)]
    r1 = np.random.Generator(np.random.PCG64(seed + 1))
    xy2 = r1.uniform(0, L, (n, 2))
    tgt0 = np.c_[xy2, _surface(xy2[:, 0], xy2[:, 1], phases, roughness)]
    # piecewise-rigid field on square blocks of side 4 x resolution
    r2 = np.random.Generator(np.random.PCG64(seed + 2))
    nb = int(np.ceil(cells / 4.0))
    B = nb * nb
    ang = np.deg2rad(r2.uniform(0, 0.5, B))
    R = _rodrigues(r2.normal(size=(B, 3)), ang)
    stable = r2.uniform(size=B) < 0.7
    t_small = r2.uniform(-0.05, 0.05, (B, 3))
    t_big = r2.uniform(0.2, 0.5, (B, 3)) * np.where(r2.uniform(size=(B, 3)) < 0.5, -1.0, 1.0)
    t = np.where(stable[:, None], t_small, t_big)
    bs = 4.0 * resolution
    bx = np.minimum((tgt0[:, 0] / bs).astype(np.int64), nb - 1)
    by = np.minimum((tgt0[:, 1] / bs).astype(np.int64), nb - 1)
    bid = by * nb + bx
    centre = np.c_[(bx + 0.5) * bs, (by + 0.5) * bs, np.zeros(n)]
    tgt = np.einsum("nij,nj->ni", R[bid], tgt0 - centre) + centre + t[bid]
    tgt[:, 2] += r1.normal(0, noise, n)
    off = np.asarray(origin, dtype=np.float64)
    return dict(src=(src + off).astype(np.float32), tgt=(tgt + off).astype(np.float32), L=L, block_R=R, block_t=t,
                block_of_tgt=bid, origin=off)


def grid_partition(pts, cells, resolution, origin=(0.0, 0.0, 0.0)):
    """(x, y) grid-cell partition -> (order (n,) int64, off (cells^2 + 1,) int64): points of patch c are
    pts[order[off[c]:off[c+1]]]. Points outside [0, L)^2 are clamped into the border cells."""
    o = np.asarray(origin, dtype=np.float64)
    cx = np.clip(((pts[:, 0].astype(np.float64) - o[0]) / resolution).astype(np.int64), 0, cells - 1)
    cy = np.clip(((pts[:, 1].astype(np.float64) - o[1]) / resolution).astype(np.int64), 0, cells - 1)
    cid = cy * cells + cx
    order = np.argsort(cid, kind="stable")
    counts = np.bincount(cid, minlength=cells * cells)
    off = np.zeros(cells * cells + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    return order, off


def _order_inside_patches(pts, order, off, cell):
    """Re-order the points of every patch along a coarse (x, y) raster of `cell`-sized bins (stable)."""
    p = pts[order].astype(np.float64)
    pid = np.repeat(np.arange(len(off) - 1), np.diff(off))
    bx = np.floor(p[:, 0] / cell).astype(np.int64)
    by = np.floor(p[:, 1] / cell).astype(np.int64)
    key = np.lexsort((bx, by, pid))
    return order[key]


def make_patches(n, cells, resolution, seed=0, noise=0.005, origin=(0.0, 0.0, 0.0), roughness=0.0, raster=None):
    """Two-epoch cloud already grouped into patch-contiguous CSR arrays.

    raster: if set, points inside a patch are ordered along an (x, y) raster of that bin size (scan-like order);
    default is the arbitrary order the sampling produced.
    Returns dict(src, src_off, tgt, tgt_off, P, max_src, max_tgt, meta)."""
    c = two_epoch_cloud(n, cells, resolution, noise=noise, seed=seed, origin=origin, roughness=roughness)
    so, soff = grid_partition(c["src"], cells, resolution, origin)
    to, toff = grid_partition(c["tgt"], cells, resolution, origin)
    if raster:
        so = _order_inside_patches(c["src"], so, soff, raster)
        to = _order_inside_patches(c["tgt"], to, toff, raster)
    return dict(src=np.ascontiguousarray(c["src"][so]), src_off=soff, tgt=np.ascontiguousarray(c["tgt"][to]), tgt_off=toff,
                P=cells * cells, max_src=int(np.diff(soff).max()), max_tgt=int(np.diff(toff).max()), meta=c)


def correspondences_from_nn(src, src_off, tgt, tgt_off, nn):
    """Kabsch-init correspondences: rows (s_i, t_nn[i]) for every source point with nn[i] >= 0, as CSR over patches.
    `nn` holds the index INSIDE the target patch (output of engine.nn_refine / oracle.nn_within). numpy in/out."""
    P = src_off.shape[0] - 1
    pid = np.repeat(np.arange(P), np.diff(src_off))
    keep = nn >= 0
    cs = src[keep]
    ct = tgt[tgt_off[pid[keep]] + nn[keep]]
    counts = np.bincount(pid[keep], minlength=P)
    off = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    return np.ascontiguousarray(cs), np.ascontiguousarray(ct), off


# ----------------------------------------------------------------------------------------------------------------------
# The same cloud model generated with torch ops ON THE DEVICE (bench.py and the full-size tests: 50 M points per epoch
# take ~150 s in numpy on one host core, < 1 s here, and every rank of a multi-GPU run can build the SAME cloud on its
# own GPU without replicating gigabytes on the host).  Point coordinates and noise come from a counter-based generator
# (splitmix64 of seed / stream / point index, integer arithmetic: identical on every rank and every device), the surface
# phases and the block motion field from the same PCG64 streams as `two_epoch_cloud`.  Same distribution, not the same
# sample, as the numpy generator (which stays the generator of the small oracle-checked test clouds).
_M64 = (1 << 64)
_GAMMA = 0x9E3779B97F4A7C15 - _M64
_MIX1 = 0xBF58476D1CE4E5B9 - _M64
_MIX2 = 0x94D049BB133111EB - _M64


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _uniform01(torch, seed, stream, idx):
    """Uniform double in [0, 1) for every int64 counter in `idx` (splitmix64 finaliser; wrapping int64 arithmetic)."""
    x = idx * _GAMMA + ((int(seed) * 1000003 + int(stream)) * 0x632BE59BD9B4E019 % _M64 - (_M64 >> 1))
    x = (x ^ _lsr(x, 30)) * _MIX1
    x = (x ^ _lsr(x, 27)) * _MIX2
    x = x ^ _lsr(x, 31)
    return _lsr(x, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def _normal(torch, seed, stream, idx):
    u1 = 1.0 - _uniform01(torch, seed, 2 * stream, idx)  # (0, 1]
    u2 = _uniform01(torch, seed, 2 * stream + 1, idx)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)


def _surface_t(torch, x, y, phases):
    z = torch.zeros_like(x)
    for j in range(1, 5):
        a, f = 2.0 / 2 ** j, 2 ** j / 100.0
        z += a * torch.sin(2 * np.pi * f * x + float(phases[j - 1, 0])) * torch.sin(2 * np.pi * f * y + float(phases[j - 1, 1]))
    return z


class _DeviceCloud:
    """The two-epoch cloud of `make_patches_device`, chunk by chunk: `chunk(lo, hi)` -> (src (m, 3), tgt (m, 3)) float32 of the points
    with generation indices lo .. hi - 1 (a pure function of the indices: any rank can produce any part)."""

    def __init__(self, cells, resolution, device, seed, noise):
        import torch
        self.torch, self.device, self.seed, self.noise = torch, device, seed, noise
        self.cells, self.resolution, self.L = cells, resolution, cells * resolution
        self.phases = np.random.Generator(np.random.PCG64(seed)).uniform(0, 2 * np.pi, (4, 2))
        r2 = np.random.Generator(np.random.PCG64(seed + 2))
        self.nb = nb = int(np.ceil(cells / 4.0))
        B = nb * nb
        ang = np.deg2rad(r2.uniform(0, 0.5, B))
        Rb = _rodrigues(r2.normal(size=(B, 3)), ang)
        stable = r2.uniform(size=B) < 0.7
        t_small = r2.uniform(-0.05, 0.05, (B, 3))
        t_big = r2.uniform(0.2, 0.5, (B, 3)) * np.where(r2.uniform(size=(B, 3)) < 0.5, -1.0, 1.0)
        tb = np.where(stable[:, None], t_small, t_big)
        self.Rb = torch.from_numpy(Rb.reshape(B, 9)).to(device)
        self.tb = torch.from_numpy(tb).to(device)

    def chunk(self, lo, hi):
        torch, seed, noise, L, nb = self.torch, self.seed, self.noise, self.L, self.nb
        bs = 4.0 * self.resolution
        idx = torch.arange(lo, hi, dtype=torch.int64, device=self.device)
        src = torch.empty((hi - lo, 3), dtype=torch.float32, device=self.device)
        tgt = torch.empty((hi - lo, 3), dtype=torch.float32, device=self.device)
        x, y = L * _uniform01(torch, seed, 0, idx), L * _uniform01(torch, seed, 1, idx)
        src[:, 0], src[:, 1] = x.to(torch.float32), y.to(torch.float32)
        src[:, 2] = (_surface_t(torch, x, y, self.phases) + noise * _normal(torch, seed, 1, idx)).to(torch.float32)
        x, y = L * _uniform01(torch, seed, 6, idx), L * _uniform01(torch, seed, 7, idx)
        z = _surface_t(torch, x, y, self.phases)
        bx = torch.clamp((x / bs).to(torch.int64), max=nb - 1)
        by = torch.clamp((y / bs).to(torch.int64), max=nb - 1)
        bid = by * nb + bx
        px, py, pz = x - (bx.to(torch.float64) + 0.5) * bs, y - (by.to(torch.float64) + 0.5) * bs, z
        R = self.Rb[bid]
        t = self.tb[bid]
        qx = R[:, 0] * px + R[:, 1] * py + R[:, 2] * pz + (bx.to(torch.float64) + 0.5) * bs + t[:, 0]
        qy = R[:, 3] * px + R[:, 4] * py + R[:, 5] * pz + (by.to(torch.float64) + 0.5) * bs + t[:, 1]
        qz = R[:, 6] * px + R[:, 7] * py + R[:, 8] * pz + t[:, 2] + noise * _normal(torch, seed, 4, idx)
        tgt[:, 0], tgt[:, 1], tgt[:, 2] = qx.to(torch.float32), qy.to(torch.float32), qz.to(torch.float32)
        return src, tgt

    def cell_of(self, p):
        torch = self.torch
        cx = torch.clamp((p[:, 0].to(torch.float64) / self.resolution).to(torch.int64), 0, self.cells - 1)
        cy = torch.clamp((p[:, 1].to(torch.float64) / self.resolution).to(torch.int64), 0, self.cells - 1)
        return cy * self.cells + cx


def make_patches_device(n, cells, resolution, device, seed=0, noise=0.005, chunk=8_000_000):
    """`make_patches` on the device: dict(src, src_off, tgt, tgt_off (torch tensors on `device`), P, max_src, max_tgt, L).
    src/tgt are float32 (n, 3) patch-contiguous, offsets int64; patch = (x, y) grid cell at `resolution`, points inside a
    patch in ascending generation index."""
    import torch

    gen = _DeviceCloud(cells, resolution, device, seed, noise)
    src = torch.empty((n, 3), dtype=torch.float32, device=device)
    tgt = torch.empty((n, 3), dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        src[lo:hi], tgt[lo:hi] = gen.chunk(lo, hi)

    def partition(p):
        cid = gen.cell_of(p)
        _, order = torch.sort(cid, stable=True)
        off = torch.zeros(cells * cells + 1, dtype=torch.int64, device=device)
        off[1:] = torch.cumsum(torch.bincount(cid, minlength=cells * cells), 0)
        return p[order].contiguous(), off

    src, soff = partition(src)
    tgt, toff = partition(tgt)
    return dict(src=src, src_off=soff, tgt=tgt, tgt_off=toff, P=cells * cells, L=gen.L,
                max_src=int((soff[1:] - soff[:-1]).max().item()), max_tgt=int((toff[1:] - toff[:-1]).max().item()))


def make_rank_share_device(n, cells, resolution, device, rank, world, seed=0, noise=0.005, chunk=8_000_000, dist=None):
    """The share of `make_patches_device`'s cloud that rank `rank` of `world` owns under the LPT assignment of
    sharding.shard_cloud -- the same dict, bit for bit, as `shard_cloud(make_patches_device(...), rank, world)` -- WITHOUT any rank
    ever holding the whole cloud: a first pass over the generator counts the points per patch (both epochs; the assignment needs
    nothing else), a second pass keeps the rank's own points.  With `dist` (an initialised torch.distributed of `world` ranks)
    the counting pass is SHARED: rank r counts every world-th chunk and one all_reduce(SUM) of the 2 P counts gives every rank
    the whole cloud's -- one pass over the cloud per node instead of one per rank (integer sums: the same counts in any order).
    Returns (share dict, ids_per_rank)."""
    import torch

    from .sharding import lpt_assign
    gen = _DeviceCloud(cells, resolution, device, seed, noise)
    P = cells * cells
    cnt = torch.zeros(2 * P, dtype=torch.int64, device=device)
    shared = dist is not None and world > 1
    for ci, lo in enumerate(range(0, n, chunk)):
        if shared and ci % world != rank:
            continue
        s, t = gen.chunk(lo, min(n, lo + chunk))
        cnt[:P] += torch.bincount(gen.cell_of(s), minlength=P)
        cnt[P:] += torch.bincount(gen.cell_of(t), minlength=P)
    if shared:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    cnt_s, cnt_t = cnt[:P], cnt[P:]
    cs, ct = cnt_s.cpu().numpy(), cnt_t.cpu().numpy()
    ids_per_rank = lpt_assign(cs.astype(np.float64) * np.maximum(ct, 1), world)  # (sharding.patch_costs on the counts)
    mine = ids_per_rank[rank]
    own = torch.zeros(P, dtype=torch.bool, device=device)
    own[torch.from_numpy(mine).to(device)] = True
    kept = {"s": ([], []), "t": ([], [])}
    for lo in range(0, n, chunk):
        s, t = gen.chunk(lo, min(n, lo + chunk))
        for key, p in (("s", s), ("t", t)):
            cid = gen.cell_of(p)
            m = own[cid]
            kept[key][0].append(p[m])
            kept[key][1].append(cid[m])
    local = torch.full((P,), -1, dtype=torch.int64, device=device)  # patch id -> place among the rank's patches (ascending ids)
    local[torch.from_numpy(mine).to(device)] = torch.arange(len(mine), device=device)

    def pack(pts, cids):
        pts, lid = torch.cat(pts), local[torch.cat(cids)]
        _, order = torch.sort(lid, stable=True)  # (stable: generation order inside a patch, as in the whole cloud)
        off = torch.zeros(len(mine) + 1, dtype=torch.int64, device=device)
        off[1:] = torch.cumsum(torch.bincount(lid, minlength=len(mine)), 0)
        return pts[order].contiguous(), off

    src, soff = pack(*kept["s"])
    tgt, toff = pack(*kept["t"])
    mx = lambda o: int((o[1:] - o[:-1]).max().item()) if len(mine) else 0  # noqa: E731
    return dict(src=src, src_off=soff, tgt=tgt, tgt_off=toff, P=len(mine), max_src=mx(soff), max_tgt=mx(toff), n_src=int(soff[-1].item())), ids_per_rank


def correspondences_from_nn_device(src, src_off, tgt, tgt_off, nn):
    """`correspondences_from_nn` with torch tensors on any device."""
    import torch
    P = src_off.shape[0] - 1
    cnt = src_off[1:] - src_off[:-1]
    pid = torch.repeat_interleave(torch.arange(P, dtype=torch.int64, device=src.device), cnt)
    keep = nn >= 0
    cs = src[keep]
    ct = tgt[tgt_off[pid[keep]] + nn[keep].to(torch.int64)]
    off = torch.zeros(P + 1, dtype=torch.int64, device=src.device)
    off[1:] = torch.cumsum(torch.bincount(pid[keep], minlength=P), 0)
    return cs.contiguous(), ct.contiguous(), off
