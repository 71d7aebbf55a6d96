#!/bin/bash
mkdir -p gpurun_out/r6y
for c in C2_1M_2k C3_10M_20k C1_50k_64; do timeout -k 10 200 python tools/gpu/p2pl_phases.py $c 2>&1 | grep -v amdgpu | tee -a gpurun_out/r6y/p2pl_other_configs.log; done
