#!/usr/bin/env python3
"""Times the reference's OWN supervoxel templates (oracle/_ref/libf4l_ref.so = the header-only codelibrary driven by
oracle/ref_harness.cpp) and the C restatement (oracle/f4l_oracle.c) on the same sample, in the BUILD CONTAINER (nothing built
from /root/reference travels to the GPU box).  The figures go into BASELINE.md section 2; bench.py's
`cpu_baseline_supervoxel` times the restatement ("port") on the GPU box's host."""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from fusion4landslide_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
c = synthetic.two_epoch_cloud(1_000_000, 45, 1.386, seed=0)
order, _ = synthetic.grid_partition(c["src"], 45, 1.386)
xyz = np.ascontiguousarray(c["src"][order][:n])  # patch-contiguous prefix: a compact strip of the tile
for name, fn in (("reference templates (oracle/_ref)", O.ref_supervoxel), ("C restatement (oracle/f4l_oracle.c)", O.supervoxel)):
    if name.startswith("reference") and not O.have_ref():
        print("oracle/_ref missing: run `make -C oracle` in the build container")
        continue
    t = time.perf_counter()
    r = fn(xyz, 30, 1.386)
    dt = time.perf_counter() - t
    print(f"{name}: {n} points, k=30, resolution 1.386 m: {dt:.2f} s = {n / dt / 1e3:.1f} k points/s, K = {r['n_supervoxels']}, 1 thread")
