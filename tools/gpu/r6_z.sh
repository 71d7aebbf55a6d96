#!/bin/bash
# round 6, call z: the headline step under the 128-register shapes (VERDICT r5 item 2's alternative): icp_kernel<0, 2, double, false> and <0, 4, double, false>
mkdir -p gpurun_out/r6z
for e in "" "F4L_ICP_NOWIDE=1" "F4L_ICP_NOWIDE=1 F4L_ICP_WAVES=4"; do
  echo "== env: ${e:-default (icp_kernel<0,2,double,true>: 168 registers, 3 waves per SIMD)}" | tee -a gpurun_out/r6z/c4_128_vgpr_shapes.log
  env $e timeout -k 10 200 python bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'kernel_ms', d['roofline'].get('kernel_ms'))" | tee -a gpurun_out/r6z/c4_128_vgpr_shapes.log
done
