#!/bin/bash
# round 6, call aq: patch normals with a window of candidates per wave -- bits against the kernel before, tests, time (window / whole patch / lane kernel)
mkdir -p gpurun_out/r6aq
timeout -k 10 300 python tools/gpu/patch_normals_ab.py pn_before > gpurun_out/r6aq/bits.log 2>&1; echo "bits rc=$?" >> gpurun_out/r6aq/bits.log
grep -v amdgpu gpurun_out/r6aq/bits.log | tail -8
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "normals or plane" 2>&1 | tail -3 | tee gpurun_out/r6aq/tests.log
for w in 1 2 0; do echo "== F4L_PATCH_NORMALS_WINDOW=$w" | tee -a gpurun_out/r6aq/time.log; F4L_PATCH_NORMALS_WINDOW=$w timeout -k 10 200 python tools/gpu/time_patch_normals.py C4_50M_100k C2_1M_2k C3_10M_20k 2>&1 | grep -v amdgpu | tee -a gpurun_out/r6aq/time.log; done
