"""Nearest-neighbour certificates on C3-like patches, simulated on the CPU (numpy + scipy.cKDTree): how many source points must
be searched again per ICP pass when a point keeps the K nearest targets of its last search and the distance every other target
keeps.  K = 1 is what icp_kernel does.  A search with bound b sees everything within b, so the
distance the others keep is min((K + 1)-th distance, b) with b = the nearest member's distance + mu.

    python tools/c3_certificate_sim.py        # prints, per (K, mu): searched points per point and pass, and those beyond one cell
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion4landslide_amd import synthetic
res=0.1; cells=16; n=int(503*cells*cells)
m=synthetic.make_patches(n,cells,res,seed=3)
P=m["P"]; R=0.1; RS=R*1.0625
def kabsch(a,b):
    ca,cb=a.mean(0),b.mean(0); H=(a-ca).T@(b-cb); U,S,Vt=np.linalg.svd(H); d=np.sign(np.linalg.det(Vt.T@U.T)); D=np.diag([1,1,d]); Rm=Vt.T@D@U.T; return Rm, cb-Rm@ca
h=R*1.0078125/8
KMAX=9
variants=[(1,h/4),(1,h/2),(2,h/4),(2,h/2),(4,h/4),(4,h/2),(4,h),(8,h/2),(8,h)]
tot=np.zeros((len(variants),21,3)); npts=0; changed=np.zeros(21)
for p in range(P):
    s=m["src"][m["src_off"][p]:m["src_off"][p+1]].astype(np.float64); t=m["tgt"][m["tgt_off"][p]:m["tgt_off"][p+1]].astype(np.float64)
    if len(t)<KMAX+1 or len(s)<10: continue
    tree=cKDTree(t)
    d,i=tree.query(s,k=1,distance_upper_bound=2*R)
    ok=np.isfinite(d)
    if ok.sum()<3: continue
    Rm,tv=kabsch(s[ok],t[i[ok]])
    cur=s@Rm.T+tv
    npts+=len(s)
    st=[None]*len(variants)
    prevnn=None
    for it in range(21):
        dk,ik=tree.query(cur,k=KMAX+1)
        nn=ik[:,0]; d1=dk[:,0]
        if prevnn is not None: changed[it]+=(nn!=prevnn).sum()
        prevnn=nn
        for v,(K,mu) in enumerate(variants):
            if it==0:
                search=np.ones(len(s),bool)
                bound=np.full(len(s),RS)
                st[v]=dict(pos=cur.copy(),M=np.zeros(len(s)),S=np.zeros((len(s),K),int))
            else:
                q=st[v]
                moved=np.linalg.norm(cur-q["pos"],axis=1)
                dS=np.linalg.norm(cur[:,None,:]-t[q["S"]],axis=2)
                dmin=dS.min(1)
                search=~(dmin<q["M"]-moved)
                bound=np.minimum(dmin+mu,RS)
            q=st[v]
            # a search with bound b sees everything within b: the K nearest among those, M = min((K+1)th, b); missing slots: repeat the nearest
            Sn=ik[:,:K].copy(); dn=dk[:,:K]
            beyond=dn>bound[:,None]
            Sn[beyond]=np.broadcast_to(ik[:,:1],Sn.shape)[beyond]
            Mn=np.minimum(dk[:,K],bound)
            q["pos"][search]=cur[search]; q["M"][search]=Mn[search]; q["S"][search]=Sn[search]
            tot[v,it]+= [search.sum(), (search&(bound>h)).sum(), 0]
        hit=d1<R
        if hit.sum()<3: break
        Ru,tu=kabsch(cur[hit],t[nn[hit]])
        cur=cur@Ru.T+tu
np.set_printoptions(suppress=True,linewidth=200,precision=3)
print("NN actually changed per pass (fraction):", (changed/npts).round(3))
for v,(K,mu) in enumerate(variants):
    a=tot[v,1:].sum(0)/npts/20
    print(f"K={K} mu={mu*1e3:.1f}mm: searched/pass {a[0]:.3f} wide {a[1]:.4f}   per pass:", (tot[v,:,0]/npts).round(2)[:12])
