"""Largest closure (visited set of one representative's turn) of the exact device segmentation on clouds of several kinds:
what the LDS queue of csrc/supervoxel_exact.hip has to hold (F4L_SV_EXACT_DEBUG prints it; a closure beyond it falls back to the
host replay).  Usage (GPU box): F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_closures.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_supervoxel_exact import _cloud
cases = [("surface", 300_000, 30, 0.5), ("surface", 300_000, 30, 2.5), ("volume", 200_000, 30, 0.5), ("volume", 200_000, 30, 1.5), ("rows", 300_000, 30, 0.6),
         ("lattice", 90_000, 30, 0.7), ("georef", 200_000, 30, 1.0), ("surface", 200_000, 60, 1.0), ("volume", 100_000, 8, 1.0)]
for kind, n, k, res in cases:
    xyz = torch.from_numpy(_cloud(kind, n, seed=1)).cuda()
    print(f"== {kind} n={n} k={k} res={res}", flush=True)
    lab, K = engine.supervoxel(xyz, k, res)
    print(f"   K={K}", flush=True)
d = synthetic.make_patches_device(1_000_000, 45, 1.386, torch.device("cuda"), seed=0)
for res in (1.386, 0.52, 5.0):
    print(f"== synthetic tile 1 M res={res}", flush=True)
    lab, K = engine.supervoxel(d["src"], 30, res)
    print(f"   K={K}", flush=True)
