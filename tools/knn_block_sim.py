"""How many candidates a lane of the kNN lane kernel measures (csrc/knn.hip, knn_lanes_kernel), simulated on the CPU: 64 consecutive
points of the cell-sorted 1 M-point tile share the block of cells their queries span plus one cell on either side (3 rows of
cells); with the wave split into 2 / 4 groups of consecutive points every group has a block of its own and the wave walks as
long as its longest group.  Prints the mean candidates per lane for the three shapes (324 / 242 / 201 when this was run).

    python tools/knn_block_sim.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion4landslide_amd import synthetic
c=synthetic.two_epoch_cloud(1_000_000,45,1.386)
p=c["src"].astype(np.float64)
n=len(p); k=30
mn=p.min(0); ext=p.max(0)-mn
target=15.0
area=np.sort(ext)[2]*np.sort(ext)[1]
h=np.sqrt(target*area/n)
for it in range(5):
    nx,ny=int(ext[0]/h)+1,int(ext[1]/h)+1
    cx=((p[:,0]-mn[0])/h).astype(np.int64); cy=((p[:,1]-mn[1])/h).astype(np.int64)
    key=cy*nx+cx
    M=len(np.unique(key)); occ=n/M
    if 0.6*target<=occ<=1.7*target: break
    h*=np.clip((target/occ)**(1/2.5),0.25,4)
print("h",h,"occ",occ,"nx,ny",nx,ny)
order=np.argsort(key,kind="stable"); key=key[order]; cx=cx[order]; cy=cy[order]
cnt=np.bincount(key,minlength=nx*ny).reshape(ny,nx)
pre=np.zeros((ny,nx+1),np.int64); pre[:,1:]=np.cumsum(cnt,1)
def block(y,x0,x1):
    tot=0
    for yy in (y-1,y,y+1):
        if 0<=yy<ny: tot+=pre[yy,min(x1+2,nx)]-pre[yy,max(x0-1,0)]
    return tot
def turns(lo,hi):
    # queries lo..hi-1 (sorted positions): one turn per row among them; returns total candidates streamed
    t=0; i=lo
    while i<hi:
        y=cy[i]; j=i
        while j<hi and cy[j]==y: j+=1
        t+=block(y,cx[i],cx[j-1]); i=j
    return t
rng=np.random.default_rng(0)
waves=rng.choice(n//64,4000,replace=False)
full=[];half=[];quart=[]
for w in waves:
    a=w*64
    full.append(turns(a,a+64))
    half.append(max(turns(a,a+32),turns(a+32,a+64)))
    quart.append(max(turns(a+16*i,a+16*i+16) for i in range(4)))
print("candidates per lane: wave64",np.mean(full),"half-wave (max of two)",np.mean(half),"quarter (max of 4)",np.mean(quart))
