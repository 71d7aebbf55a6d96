import os, sys
import numpy as np, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from fusion4landslide_amd import engine as eng, synthetic
from oracle import oracle as O
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def disp(d,T,Tr):
    out=[]
    for p in range(d["P"]):
        s = d["src"][d["src_off"][p]:d["src_off"][p+1]].astype(np.float64)
        if len(s)==0: out.append(0); continue
        out.append(np.abs((s@T[p,:3,:3].T+T[p,:3,3])-(s@Tr[p,:3,:3].T+Tr[p,:3,3])).max())
    return np.array(out)

print("=== kNN d2 diag")
g = np.load(os.path.join(ROOT, "tests/golden/supervoxel_surf_s0_n2000_k15.npz"))
xyz, k = g["xyz"], int(g["k"])
idx, d2 = eng.knn(dev(xyz), k, return_d2=True); idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
print("entries differ", (d2 != g["knn_d2"]).sum(), "idx equal frac", (idx==g["knn_idx"]).mean())

print("=== ICP p2p diag")
for origin in [(0.,0.,0.),(2647.,1177.,1500.)]:
    d = synthetic.make_patches(30000, 6, 1.386, seed=1, origin=origin)
    for fixed,mi in [(True,20),(False,30)]:
        ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=mi, fixed_iters=fixed)
        for search in ("f32","f64"):
            out = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1, max_iter=mi, fixed_iters=fixed, search=search)
            dd = disp(d, out["T"].cpu().numpy(), ref["T"])
            print(origin, fixed, search, "max %.2e med %.2e p95 %.2e n>1e-4 %d"%(dd.max(), np.median(dd), np.percentile(dd,95), (dd>1e-4).sum()), "fit %.2e rmse %.2e"%(np.abs(out["fitness"].cpu().numpy()-ref["fitness"]).max(), np.abs(out["rmse"].cpu().numpy()-ref["rmse"]).max()), "iters eq", (out["iters"].cpu().numpy()==ref["iters"]).mean())

print("=== p2plane diag (rough surface)")
for rough in (0.0, 0.15):
    d = synthetic.make_patches(24000, 5, 1.386, seed=4, roughness=rough)
    nrm = eng.patch_normals(dev(d["tgt"]), dev(d["tgt_off"]), 30)
    for it in (1,5,30):
        ref = O.piecewise_icp(d["src"], d["src_off"], d["tgt"], d["tgt_off"], max_corr_dist=0.1, max_iter=it, fixed_iters=True, icp_type="point2plane")
        for search in ("f32","f64"):
            out = eng.piecewise_icp(dev(d["src"]), dev(d["src_off"]), dev(d["tgt"]), dev(d["tgt_off"]), max_corr_dist=0.1, max_iter=it, fixed_iters=True, icp_type="point2plane", tgt_normals=nrm, search=search)
            dd = disp(d, out["T"].cpu().numpy(), ref["T"])
            print("rough",rough,"iters",it,search,"max %.2e med %.2e"%(dd.max(),np.median(dd)),"argmax",dd.argmax(), "rmse diff %.2e"%np.abs(out["rmse"].cpu().numpy()-ref["rmse"]).max())

print("=== edge: single point patch")
src=np.array([[0.1,0.2,0.3]],np.float32); tgt=src+np.float32(0.01)
out=eng.piecewise_icp(dev(src),dev(np.array([0,1])),dev(tgt),dev(np.array([0,1])),max_corr_dist=0.1)
print(out["T"].cpu().numpy()[0], O.icp(src,tgt)["est_transform"])
