#!/bin/bash
mkdir -p gpurun_out/r6w
timeout -k 10 300 python tools/gpu/time_point2plane.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6w/p2pl.log
