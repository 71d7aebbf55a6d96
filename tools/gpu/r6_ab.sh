#!/bin/bash
# round 6, call ab: the bulk shapes in two translation units (default scheduler / iterative-minreg) -- tests, times
mkdir -p gpurun_out/r6ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mirrors.py tests/test_gpu_fine_matching.py -m gpu -x -q -k "icp or plane or loop" 2>&1 | tail -4 | tee gpurun_out/r6ab/tests.log
timeout -k 10 300 python tools/gpu/p2pl_phases.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6ab/p2pl_time.log
timeout -k 10 300 python bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r6ab/bench_c4.log
for c in C2_1M_2k C3_10M_20k; do timeout -k 10 300 python bench.py --config $c --cpu-seconds 0 --extras 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/r6ab/bench_c4.log; done
