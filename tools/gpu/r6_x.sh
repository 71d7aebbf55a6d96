#!/bin/bash
# round 6, call x: point-to-plane with the sums in a loop of their own, totals through LDS -- tests, time (product and the 256-register build)
mkdir -p gpurun_out/r6x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mirrors.py -m gpu -x -q -k "icp or plane" 2>&1 | tail -4 | tee gpurun_out/r6x/tests.log
TAIL=3 timeout -k 10 600 bash tools/gpu/lib_ab.sh "timeout -k 10 200 python tools/gpu/p2pl_phases.py" icp_plane_wpe2 2>&1 | tee gpurun_out/r6x/p2pl_time.log
