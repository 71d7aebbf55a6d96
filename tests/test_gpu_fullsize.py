"""GPU tests at BASELINE.json's full sizes (configs[2] "C3_10M_20k", configs[3] "C4_50M_100k", configs[4] "C5_100M_full"): the oracle cannot run whole
clouds of that size in seconds, so each test checks size-independent properties of the whole result and a bounded sample
of patches (>= 24, the largest patch among them) against the CPU oracle in float64 mode, through the same fused launch
`bench.py` times (f4l_patch_loop: Kabsch init -> 20 fixed ICP iterations -> displacement rows).

Tolerances (float64 search = the reference's arithmetic): displacement of every patch point under the two transforms
<= 1e-9 m, fitness equal to 1e-12, rmse to 1e-10 -- the figures of tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

MAX_CORR, MAX_ITER = 0.1, 20


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from fusion4landslide_amd import engine
    return engine


def _problem(eng, name, seed=0):
    from fusion4landslide_amd import synthetic
    c = synthetic.CONFIGS[name]
    dev = torch.device("cuda")
    d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], dev, seed=seed)
    P = d["P"]
    eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(P, 1, 1)
    nn, _ = eng.nn_refine(d["src"], d["src_off"], d["tgt"], d["tgt_off"], eye,
                          torch.full((P,), 2 * MAX_CORR, dtype=torch.float64, device=dev), max_tgt_patch=d["max_tgt"],
                          return_rows=False)
    cs, ct, coff = synthetic.correspondences_from_nn_device(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn)
    return d, cs, ct, coff, nn


def _step(eng, d, cs, ct, coff, **kw):
    return eng.patch_loop(d["src"], d["src_off"], d["tgt"], d["tgt_off"], cs, ct, coff, None, 0.0, 1e-6, max_corr_dist=MAX_CORR,
                          max_iter=MAX_ITER, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], **kw)


def _check_properties(d, out, n):
    T = out["T"]
    assert bool((out["iters"] == MAX_ITER).all())
    R = T[:, :3, :3]
    assert float((R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device=T.device)).abs().max()) < 1e-9
    assert float((torch.linalg.det(R) - 1.0).abs().max()) < 1e-9
    assert bool((T[:, 3, :] == torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64, device=T.device)).all())
    fit, rmse = out["fitness"], out["rmse"]
    assert bool(((fit >= 0) & (fit <= 1)).all()) and bool((rmse <= MAX_CORR).all()) and bool(torch.isfinite(T).all())
    rows = out["rows"]
    assert rows.shape == (n, 6) and torch.equal(rows[:, :3], d["src"])
    # rows[:, 3:] is T_p s for every point (checked in double on the device against the returned transforms)
    cnt = d["src_off"][1:] - d["src_off"][:-1]
    pid = torch.repeat_interleave(torch.arange(d["P"], device=T.device), cnt, output_size=n)
    lo = 0
    worst = 0.0
    while lo < n:  # chunked: the (n, 3, 3) gather of a 50 M cloud would be 3.6 GB at once
        hi = min(n, lo + 5_000_000)
        Tp = T[pid[lo:hi]]
        s = d["src"][lo:hi].to(torch.float64)
        q = torch.einsum("nij,nj->ni", Tp[:, :3, :3], s) + Tp[:, :3, 3]
        worst = max(worst, float((q - rows[lo:hi, 3:].to(torch.float64)).abs().max()))
        lo = hi
    assert worst <= 2e-5, worst  # float32 rounding of coordinates up to ~440 m (C4); the arithmetic is double
    return fit


def _check_sample_against_oracle(d, cs, ct, coff, out, pick):
    so, to, co = d["src_off"].cpu().numpy(), d["tgt_off"].cpu().numpy(), coff.cpu().numpy()
    T, fit, rmse = out["T"].cpu().numpy(), out["fitness"].cpu().numpy(), out["rmse"].cpu().numpy()
    worst = 0.0
    for p in pick:
        s = d["src"][so[p]:so[p + 1]].cpu().numpy()
        t = d["tgt"][to[p]:to[p + 1]].cpu().numpy()
        a, b = cs[co[p]:co[p + 1]].cpu().numpy(), ct[co[p]:co[p + 1]].cpu().numpy()
        T0 = np.eye(4)
        if len(a):
            R, tt = O.kabsch_batched(a, b, np.array([0, len(a)], dtype=np.int64), eps=1e-6)
            T0[:3, :3], T0[:3, 3] = R[0], tt[0]
        one = O.icp(s, t, init_T=T0, max_corr_dist=MAX_CORR, max_iter=MAX_ITER, fixed_iters=True)
        s64 = s.astype(np.float64)
        ref = s64 @ one["est_transform"][:3, :3].T + one["est_transform"][:3, 3]
        got = s64 @ T[p, :3, :3].T + T[p, :3, 3]
        dev = float(np.abs(got - ref).max()) if len(s) else 0.0
        worst = max(worst, dev)
        assert dev <= 1e-9, (p, len(s), len(t), dev)
        assert abs(fit[p] - one["fitness"]) < 1e-12 and abs(rmse[p] - one["inlier_rmse"]) < 1e-10, p
    return worst


def test_full_size_C4_50M_100k(eng):
    """BASELINE.json configs[3] / the metric's own cloud on one GPU: 50 M points per epoch, 99 856 patches."""
    d, cs, ct, coff, _ = _problem(eng, "C4_50M_100k")
    n = 50_000_000
    assert d["P"] == 316 * 316 and d["src"].shape == (n, 3) and int(d["src_off"][-1]) == n
    out = _step(eng, d, cs, ct, coff)
    fit = _check_properties(d, out, n)
    assert float(fit.mean()) > 0.5  # 70 % of the blocks are displaced by less than the radius
    rng = np.random.default_rng(0)
    size = (d["src_off"][1:] - d["src_off"][:-1]).cpu().numpy()
    pick = np.unique(np.r_[np.linspace(0, d["P"] - 1, 24).astype(int), int(size.argmax()), int(size.argmin()),
                           rng.integers(0, d["P"], 6)])
    _check_sample_against_oracle(d, cs, ct, coff, out, pick)
    # idempotence: restarting ICP from the result moves nothing beyond the tolerance on the converged patches
    again = eng.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], init_T=out["T"], max_corr_dist=MAX_CORR, max_iter=30,
                              max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"])
    moved = (again["T"][:, :3, 3] - out["T"][:, :3, 3]).abs().amax(dim=1)
    assert float(moved.median()) < 1e-4
    # run-to-run bit reproducible at full size
    out2 = _step(eng, d, cs, ct, coff, return_rows=False)
    assert torch.equal(out2["T"], out["T"]) and torch.equal(out2["rmse"], out["rmse"])


def test_sharded_C4_equals_single_launch(eng):
    """The multi-GPU split of bench.py (--gpus 2: LPT shares of ONE cloud) run share by share on this GPU: per-patch results
    scattered back to global order are bit-equal to the single launch over the whole cloud (a patch's result does not
    depend on which launch, or which workgroup, it ran in -- while the launches have the same shape: both shares and the
    whole are throughput-shape batches here, >= 6144 patches; across shapes the sums are taken in another order and the
    results agree to rounding, test_icp_throughput_shape_equals_the_latency_shape).  On a 16 M-point cloud of the C4 density."""
    from fusion4landslide_amd import sharding, synthetic
    dev = torch.device("cuda")
    cloud = synthetic.make_patches_device(16_000_000, 180, 1.386, dev, seed=5)
    P = cloud["P"]

    def run(d):
        eye = torch.eye(4, dtype=torch.float64, device=dev).repeat(d["P"], 1, 1)
        nn, _ = eng.nn_refine(d["src"], d["src_off"], d["tgt"], d["tgt_off"], eye,
                              torch.full((d["P"],), 2 * MAX_CORR, dtype=torch.float64, device=dev), max_tgt_patch=d["max_tgt"],
                              return_rows=False)
        cs, ct, coff = synthetic.correspondences_from_nn_device(d["src"], d["src_off"], d["tgt"], d["tgt_off"], nn)
        return _step(eng, d, cs, ct, coff, return_rows=False)

    whole = run(cloud)
    T = torch.zeros_like(whole["T"])
    fit = torch.zeros_like(whole["fitness"])
    seen = torch.zeros(P, dtype=torch.int32, device=dev)
    for rank in range(2):
        d, ids = sharding.shard_cloud(cloud, rank, 2)
        out = run(d)
        mine = torch.from_numpy(ids[rank]).to(dev)
        T[mine], fit[mine] = out["T"], out["fitness"]
        seen[mine] += 1
    assert bool((seen == 1).all())
    assert torch.equal(T, whole["T"]) and torch.equal(fit, whole["fitness"])


def test_full_size_C3_10M_20k_dense(eng):
    """BASELINE.json configs[2]: 10 M points in 0.1 m patches, correspondence radius = patch size (every query has
    hundreds of targets inside its radius; the border patches collect the points the motion field pushed outside and
    exceed 4096 points)."""
    d, cs, ct, coff, _ = _problem(eng, "C3_10M_20k")
    n = 10_000_000
    assert d["P"] == 141 * 141 and int(d["src_off"][-1]) == n
    size_s = (d["src_off"][1:] - d["src_off"][:-1]).cpu().numpy()
    size_t = (d["tgt_off"][1:] - d["tgt_off"][:-1]).cpu().numpy()
    assert max(size_s.max(), size_t.max()) > 4096
    out = _step(eng, d, cs, ct, coff)
    _check_properties(d, out, n)
    rng = np.random.default_rng(1)
    big = np.argsort(-np.maximum(size_s, size_t))[:3]  # the largest patches (targets beyond 4096)
    pick = np.unique(np.r_[np.linspace(0, d["P"] - 1, 24).astype(int), big, rng.integers(0, d["P"], 5)])
    _check_sample_against_oracle(d, cs, ct, coff, out, pick)


def _occupied_cells_device(xyz, res):
    """grid_sample.h:48-68 on the device (double arithmetic like the reference): the number of occupied cells of the resolution
    grid anchored at the cloud's bounding-box minimum."""
    mn = xyz.min(dim=0).values.double()
    mx = xyz.max(dim=0).values.double()
    size = ((mx - mn) / res + 1).to(torch.int64)
    key = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    for d in range(3):
        c = torch.clamp(((xyz[:, d].double() - mn[d]) / res).to(torch.int64), min=0)
        c = torch.minimum(c, size[d] - 1)
        key = key * size[d] + c
    return int(torch.unique(key).shape[0])


@pytest.mark.parametrize("partition", ["identical", "parallel"])
def test_full_size_C5_100M_full_path(eng, partition, monkeypatch):
    """BASELINE.json configs[4] on ONE GPU: the whole hot path -- median resolution, supervoxel partition on the device,
    patches, point matches, per-patch Kabsch + 20-iteration ICP + rows, nearest-neighbour refinement -- on 100 M points per
    epoch (pipeline.full_path, what bench.py's `full_path_100M_*_partition` extras time).  `identical` is the path's default and
    the entry points': the REFERENCE's supervoxel labels (supervoxel_segmentation.h:117-237 on the device, f4l_supervoxel);
    `parallel` the opt-in variant.  Size-independent properties of the whole result, >= 30 sampled patches of the per-patch stage
    against the CPU oracle to 1e-9 m, and -- for the reference's labels -- every label of a 2 M-point sub-tile against the one-core
    replay of the reference's sequence (the whole 100 M against the replay: tools/gpu/svx_100M_vs_host.py, profiles/r6_*)."""
    from fusion4landslide_amd import pipeline, synthetic
    monkeypatch.delenv("F4L_SV_EXACT_HOST", raising=False)
    assert pipeline.full_path.__defaults__[3] == "identical"  # (the default of the path = the default of main_fusion / main_piecewise_icp)
    c = synthetic.CONFIGS["C5_100M_full"]
    n = c["n"]
    dev = torch.device("cuda")
    d = synthetic.make_patches_device(n, c["cells"], c["resolution"], dev, seed=0)
    src, tgt = d["src"], d["tgt"]
    del d
    torch.cuda.empty_cache()
    r = pipeline.full_path(src, tgt, max_iter=MAX_ITER, fixed_iters=True, keep_inputs=True, partition=partition)
    K, labels = r["K"], r["labels"]
    # the partition: K = occupied cells of the resolution grid exactly (grid_sample.h:48-68), labels 0 .. K-1 all non-empty
    assert K == _occupied_cells_device(src, r["resolution"])
    cnt = torch.bincount(labels.to(torch.int64), minlength=K)
    assert cnt.shape[0] == K and int(cnt.min()) > 0 and int(labels.min()) == 0 and int(cnt.sum()) == n
    off = r["src_off"]
    assert off.shape[0] == K + 1 and int(off[-1]) == n and torch.equal(off[1:] - off[:-1], cnt)
    # the per-patch stage: every patch ran its 20 iterations; rotations orthonormal; fitness / rmse in range; rows = T s
    T = r["T"]
    assert bool((r["iters"] == MAX_ITER).all())
    R = T[:, :3, :3]
    fit, rmse = r["fitness"], r["rmse"]
    ncorr = r["corr_off"][1:] - r["corr_off"][:-1]
    ortho = (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device=dev)).abs().amax(dim=(1, 2))
    # (of 1.67 M supervoxels a handful start from a degenerate fit -- e.g. seven matches that all name the same target point:
    #  cross-covariance zero, any rotation is "the" Kabsch answer -- and keep 1e-8 of non-orthonormality; everything well posed is
    #  orthonormal to rounding)
    posed = (fit >= 0.5) & (ncorr >= 10)
    assert float(ortho[posed].max()) < 1e-12 and float(ortho.max()) < 1e-6 and float((ortho > 1e-12).double().mean()) < 1e-4
    assert float((torch.linalg.det(R) - 1.0).abs().max()) < 1e-6
    assert bool(((fit >= 0) & (fit <= 1)).all()) and bool((rmse <= MAX_CORR).all()) and bool(torch.isfinite(T).all())
    assert float(fit.mean()) > 0.5
    rows, ps = r["rows"], r["patch_src"]
    assert rows.shape == (n, 6) and torch.equal(rows[:, :3], ps) and torch.equal(ps, src[r["order"].to(torch.int64)])
    pid = torch.repeat_interleave(torch.arange(K, device=dev), cnt, output_size=n)
    worst, lo = 0.0, 0
    while lo < n:
        hi = min(n, lo + 5_000_000)
        Tp = T[pid[lo:hi]]
        q = torch.einsum("nij,nj->ni", Tp[:, :3, :3], ps[lo:hi].to(torch.float64)) + Tp[:, :3, 3]
        worst = max(worst, float((q - rows[lo:hi, 3:].to(torch.float64)).abs().max()))
        lo = hi
    assert worst <= 5e-5, worst  # float32 rounding of coordinates up to ~620 m; the arithmetic is double
    del pid
    # >= 30 sampled patches (the largest and the smallest among them) against the oracle: Kabsch init -> 20 ICP iterations
    # (supervoxels of ~60 points: the sample is drawn from the well-posed ones -- >= 30 points in both epochs, fitness >= 0.5 --;
    #  a patch of a dozen points, or one whose epochs barely overlap, may settle in another local solution after the first
    #  rounding difference, on the device and on the host alike)
    rng = np.random.default_rng(5)
    size = cnt.cpu().numpy()
    size_t = (r["tgt_off"][1:] - r["tgt_off"][:-1]).cpu().numpy()
    ok = np.nonzero((size >= 30) & (size_t >= 30) & (fit.cpu().numpy() >= 0.5) & (ncorr.cpu().numpy() >= 10))[0]
    assert len(ok) > K // 2
    pick = np.unique(np.r_[ok[np.linspace(0, len(ok) - 1, 28).astype(int)], ok[int(size[ok].argmax())], ok[int(size[ok].argmin())],
                           ok[rng.integers(0, len(ok), 8)]])
    dd = dict(src=ps, tgt=r["patch_tgt"], src_off=off, tgt_off=r["tgt_off"])
    assert len(pick) >= 30
    _check_sample_against_oracle(dd, r["corr_src"], r["corr_ref"], r["corr_off"], dict(T=T, fitness=fit, rmse=rmse), pick)
    # bit reproducible at full size: the partition and the transforms of a second run
    keep = dict(labels=labels.clone(), T=T.clone(), K=K)
    del r, rows, ps, dd, T, R
    torch.cuda.empty_cache()
    r2 = pipeline.full_path(src, tgt, max_iter=MAX_ITER, fixed_iters=True, partition=partition)
    assert r2["K"] == keep["K"] and torch.equal(r2["labels"], keep["labels"]) and torch.equal(r2["T"], keep["T"])
    print(f"C5 100 M points, {partition} partition: K = {K}, stages (ms) " + ", ".join(f"{k_} {v:.1f}" for k_, v in r2["stage_ms"].items()))
    if partition != "identical":
        return
    res = r2["resolution"]
    del r2
    eng.release_scratch()
    torch.cuda.empty_cache()
    # the reference's labels: every label of a 2 M-point sub-tile (a compact strip of the same cloud, at the run's resolution) against the one-core
    # replay of the reference's sequence (csrc/supervoxel_host.cpp, itself pinned by the reference-compiled fixtures)
    sub = src[:2_000_000].contiguous()
    lab_d, K_d = eng.supervoxel(sub, 30, res)
    monkeypatch.setenv("F4L_SV_EXACT_HOST", "1")
    lab_h, K_h = eng.supervoxel(sub, 30, res)
    assert K_d == K_h and torch.equal(lab_d, lab_h), f"{int((lab_d != lab_h).sum())} of 2 M labels differ from the host replay"
