#!/bin/bash
mkdir -p gpurun_out/r6ar
for w in 2 6 0; do echo "== F4L_PATCH_NORMALS_WINDOW=$w" | tee -a gpurun_out/r6ar/time.log; F4L_PATCH_NORMALS_WINDOW=$w timeout -k 10 200 python tools/gpu/time_patch_normals.py C4_50M_100k C2_1M_2k 2>&1 | grep -v amdgpu | tee -a gpurun_out/r6ar/time.log; done
