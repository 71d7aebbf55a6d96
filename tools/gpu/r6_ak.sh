#!/bin/bash
mkdir -p gpurun_out/r6ak
for e in "" "F4L_SV_EXACT_XCH_JACOBI=1"; do
  echo "== ${e:-in place}" | tee -a gpurun_out/r6ak/xch_passes.log
  env $e F4L_SV_EXACT_DEBUG=1 timeout -k 10 200 python tools/gpu/svx_only.py 10000000 1 2>&1 | grep "generation\|exchange passes" | tee -a gpurun_out/r6ak/xch_passes.log
done
