#!/bin/bash
# round 6, call aj: the exchange's estimate updated in place (default) against two buffers (F4L_SV_EXACT_XCH_JACOBI=1): labels, passes, time
mkdir -p gpurun_out/r6aj
timeout -k 10 300 python -m pytest tests/test_gpu_supervoxel_exact.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6aj/tests.log
grep -q "passed" gpurun_out/r6aj/tests.log && ! grep -q "failed\|error" gpurun_out/r6aj/tests.log || { echo "tests not green: stopping"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mirrors.py -m gpu -x -q -k "supervoxel" 2>&1 | tail -3 | tee -a gpurun_out/r6aj/tests.log
for n in 1000000 10000000; do
  for e in "" "F4L_SV_EXACT_XCH_JACOBI=1"; do
    echo "== $n ${e:-in place}" | tee -a gpurun_out/r6aj/xch_in_place.log
    env $e F4L_SV_EXACT_DEBUG=1 timeout -k 10 200 python tools/gpu/svx_only.py $n 3 2>&1 | grep -v amdgpu | grep "f4l_supervoxel\|exchange passes" | tail -4 | tee -a gpurun_out/r6aj/xch_in_place.log
  done
done
timeout -k 10 600 python tools/gpu/fuzz_supervoxel_exact.py 60 9000 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r6aj/fuzz_svx.log
