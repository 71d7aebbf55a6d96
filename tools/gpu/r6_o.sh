#!/bin/bash
# round 6, call o: where f4l_patch_normals_f64's time goes at C4 -- time-only ablation builds (tools/build_variant.sh pl_* XDEF=-DPL_ABL_*)
mkdir -p gpurun_out/r6o
TAIL=1 bash tools/gpu/lib_ab.sh "python tools/gpu/time_patch_normals.py C4_50M_100k" pl_nopass1 pl_nopass2 pl_nopass12 pl_nosort pl_noeig > gpurun_out/r6o/patch_normals_ablation.log 2>&1
cat gpurun_out/r6o/patch_normals_ablation.log
