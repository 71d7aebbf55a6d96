#!/bin/bash
# round 6, call p: the reworked passes of patch_normals_lanes_kernel -- bit equality with the kernel before, the parity tests, time
mkdir -p gpurun_out/r6p
python tools/gpu/patch_normals_ab.py pl_round5_seq > gpurun_out/r6p/patch_normals_bits.log 2>&1; echo "bits rc=$?" >> gpurun_out/r6p/patch_normals_bits.log
grep -v amdgpu gpurun_out/r6p/patch_normals_bits.log | tail -12
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "normals or point2plane or plane" 2>&1 | tail -4 | tee gpurun_out/r6p/tests.log
TAIL=3 bash tools/gpu/lib_ab.sh "python tools/gpu/time_patch_normals.py C4_50M_100k C2_1M_2k C3_10M_20k" pl_round5_seq 2>&1 | tee gpurun_out/r6p/patch_normals_time.log
