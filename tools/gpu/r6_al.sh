#!/bin/bash
mkdir -p gpurun_out/r6al
for n in 1000000 10000000; do F4L_SV_EXACT_DEBUG=1 timeout -k 10 400 python tools/gpu/svx_orders.py $n 2>&1 | grep -v amdgpu | grep "^==\|points,\|rounds," | awk '/^==/{h=$0; c=0; print; next} /rounds,/{c++; if(c==1) print; next} {print}' | tee -a gpurun_out/r6al/svx_orders.log; done
