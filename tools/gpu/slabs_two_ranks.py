"""The N > 1 path of the full hot path (pipeline.full_path_slabs: slab partition with halo, second epoch joined through the
slabs, per-patch stages where the patches live) with this library's HIP kernels on real devices.  On a node with >= WORLD_SIZE
GPUs: one rank per GPU over RCCL.  On a one-GPU box: all ranks on that GPU with gloo (host-staged collectives) -- the same
code path except for the transport.  Launch (nothing may have touched the GPU in the launching process):

    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/gpu/slabs_two_ranks.py [n_points]

Rank 0 also runs the whole tile alone (pipeline.full_path) and checks the split run against it."""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
multi = torch.cuda.device_count() >= world
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) if multi else 0)
dist.init_process_group("nccl" if multi else "gloo", rank=rank, world_size=world)
from fusion4landslide_amd import pipeline, synthetic

dev = torch.device("cuda")
c = synthetic.make_patches_device(n, int(round((n / 1e6) ** 0.5 * 45)), 1.386, dev, seed=0)  # the SAME cloud on every rank
src, tgt = c["src"], c["tgt"]
del c
RES, HALO = 0.526, 0.6
gid = torch.arange(rank, src.shape[0], world, device=dev)
tsel = torch.arange(world - 1 - rank, tgt.shape[0], world, device=dev)
for it in range(2):  # (second pass: warm)
    out = pipeline.full_path_slabs(src[gid].contiguous(), gid, tgt[tsel].contiguous(), dist, rank, world, HALO, RES, max_iter=20, fixed_iters=True)
print(f"[rank {rank}] owned {out['gid'].shape[0]} source points, {out['K_local']} of {out['K_total']} supervoxels, halo {out['n_halo']}, "
      f"forwarded {out['n_forwarded']} target points, uncertified {out['n_uncertified']}; stages ms "
      + " ".join(f"{k} {v:.2f}" for k, v in out["stage_ms"].items()), flush=True)
# displacement of every owned source point by global id -> rank 0
cnt = (out["src_off"][1:] - out["src_off"][:-1])
pid = torch.repeat_interleave(torch.arange(out["K_local"], device=dev), cnt)
rows = out["rows"].double()
T = out["T"][pid]
pred = torch.einsum("nij,nj->ni", T[:, :3, :3], rows[:, :3]) + T[:, :3, 3]
row_err = float((pred - rows[:, 3:]).abs().max())
assert torch.equal(rows[:, :3].float(), src[out["gid"]]), "rows must hold the owned source points in patch order"
mine = dict(gid=out["gid"].cpu(), disp=(rows[:, 3:] - rows[:, :3]).cpu(), fit=out["fitness"].cpu(), cnt=cnt.cpu(), row_err=row_err,
            pfit=out["fitness"][pid].cpu(), rmse=out["rmse"].cpu(),
            K_local=out["K_local"], K_total=out["K_total"], bad=out["n_uncertified"])
allr = [None] * world
dist.all_gather_object(allr, mine)
if rank == 0:
    whole = pipeline.full_path(src, tgt, resolution=RES, max_iter=20, fixed_iters=True, partition="parallel")
    whole = pipeline.full_path(src, tgt, resolution=RES, max_iter=20, fixed_iters=True, partition="parallel")
    wd = torch.zeros((src.shape[0], 3), dtype=torch.float64)
    wr = whole["rows"].double().cpu()
    wd[whole["order"].to(torch.int64).cpu()] = wr[:, 3:] - wr[:, :3]
    g = torch.cat([r["gid"] for r in allr])
    assert torch.equal(torch.sort(g).values, torch.arange(src.shape[0])), "every source point on exactly one rank"
    assert all(r["bad"] == 0 for r in allr) and sum(r["K_local"] for r in allr) == allr[0]["K_total"] == whole["K"]
    assert max(r["row_err"] for r in allr) < 1e-5
    sd = torch.zeros_like(wd)
    sd[g] = torch.cat([r["disp"] for r in allr])
    diff = (sd - wd).norm(dim=1)
    # The partitions differ (a slab is segmented on its own), so the patches differ, and on this smooth surface a patch's rigid
    # fit is free to slide in its own plane (the aperture problem of point-to-point ICP on a near-planar patch): single
    # points' displacements differ by centimetres between ANY two partitions.  What must agree is how well the epochs
    # register: the share of points with a correspondence and the residual of those.
    fit_s = float(sum((r["fit"] * r["cnt"]).sum() for r in allr) / src.shape[0])
    wc = (whole["src_off"][1:] - whole["src_off"][:-1]).cpu()
    fit_w = float((whole["fitness"].cpu() * wc).sum() / src.shape[0])

    def wmean(v, w):
        ok = torch.isfinite(v)
        return float((v[ok] * w[ok]).sum() / w[ok].sum())
    rm_s = wmean(torch.cat([r["rmse"] for r in allr]), torch.cat([r["fit"] * r["cnt"] for r in allr]))
    rm_w = wmean(whole["rmse"].cpu(), whole["fitness"].cpu() * wc)
    print(f"whole tile alone: K {whole['K']}, total {whole['stage_ms']['total']:.2f} ms; split over {world} ranks "
          f"({'RCCL, one GPU each' if multi else 'gloo, one shared GPU'}): same K, point-weighted fitness {fit_s:.4f} against {fit_w:.4f}, "
          f"inlier rmse {1e3 * rm_s:.3f} mm against {1e3 * rm_w:.3f} mm; displacement of a source point, split against whole: median "
          f"{float(diff.median()):.2e} m, {100 * float((diff < 1e-9).float().mean()):.1f} % identical (points in supervoxels both runs cut alike)")
    assert abs(fit_s - fit_w) < 0.01 and abs(rm_s - rm_w) < 0.05 * rm_w
    print("OK")
dist.barrier()
dist.destroy_process_group()
