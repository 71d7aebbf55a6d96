#!/bin/bash
# round 6, call ai: the fuzzers once more on the final kernels (ICP incl. point-to-plane through the new summing order and translation units; the exact partition)
mkdir -p gpurun_out/r6ai
timeout -k 10 900 python tools/gpu/fuzz_icp.py 60 7000 2>&1 | grep -v amdgpu | tail -25 | tee gpurun_out/r6ai/fuzz_icp.log
F4L_ICP_THROUGHPUT=1 timeout -k 10 900 python tools/gpu/fuzz_icp.py 30 8000 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r6ai/fuzz_icp_throughput_shapes.log
timeout -k 10 900 python tools/gpu/fuzz_supervoxel_exact.py 60 5000 2>&1 | grep -v amdgpu | tail -6 | tee gpurun_out/r6ai/fuzz_svx.log
