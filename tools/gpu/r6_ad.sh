#!/bin/bash
# round 6, call ad: supervoxel_exact.hip and patch_ops.hip under other instruction schedulers
mkdir -p gpurun_out/r6ad
TAIL=1 timeout -k 10 900 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svx_only.py 10000000 3" s_svx_maxilp s_svx_minreg 2>&1 | tee gpurun_out/r6ad/svx_schedulers.log
TAIL=1 timeout -k 10 900 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/time_patch_normals.py C4_50M_100k" s_po_maxilp s_po_minreg 2>&1 | tee gpurun_out/r6ad/patch_ops_schedulers.log
