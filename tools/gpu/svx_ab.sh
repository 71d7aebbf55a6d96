#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/svx_ab.sh <variant> ...  -- f4l_supervoxel (the reference's labels on the device) under library
# variants: the device == host-replay tests, then the time of a 1 M-point tile at two resolutions
cat > /tmp/svx_time.py <<P
import sys, time, torch
sys.path.insert(0, "$PWD")
from fusion4landslide_amd import engine, synthetic
d = synthetic.make_patches_device(1000000, 45, 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
for res in (1.386, 0.52):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); lab, K = engine.supervoxel(xyz, 30, res); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print(f"f4l_supervoxel 1 M points, resolution {res}: {min(ts[1:]):.1f} ms, K={K}", flush=True)
P
for V in product "$@"; do
  if [ "$V" = product ]; then L=""; else L="$PWD/fusion4landslide_amd/lib/variants/lib_$V.so"; fi
  echo "== $V"
  F4L_LIB_PATH=$L timeout 300 python -m pytest tests/test_gpu_supervoxel_exact.py -x -q -m gpu 2>&1 | tail -1
  F4L_LIB_PATH=$L timeout 300 python3 /tmp/svx_time.py 2>&1 | grep f4l_supervoxel
done
