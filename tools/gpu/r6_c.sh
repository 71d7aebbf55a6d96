#!/bin/bash
mkdir -p gpurun_out/r6c
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c
TAIL=3 bash tools/gpu/lib_ab.sh "python3 tools/gpu/svx_only.py 10000000 3" svx_noxcd svx_wpe6 > $O/svx_ab_10M.log 2>&1
TAIL=3 bash tools/gpu/lib_ab.sh "python3 tools/gpu/svx_only.py 1000000 3" svx_noxcd svx_wpe6 > $O/svx_ab_1M.log 2>&1
grep -E "==|f4l_supervoxel" $O/svx_ab_10M.log $O/svx_ab_1M.log | grep -v "round 1" | tail -30
python -m pytest tests/test_pcd_tiling.py tests/test_fusion_entry.py tests/test_gpu_supervoxel_exact.py tests/test_gpu_supervoxel_parallel.py -x -q 2>&1 | tail -25 > $O/tests_a.log
tail -6 $O/tests_a.log
python3 tools/gpu/time_main_fusion.py 1000000 4 > $O/main_fusion_tiles.log 2>&1
grep -E "main_fusion:|tottime" -A14 $O/main_fusion_tiles.log | head -60
F4L_ASYNC_IO=0 python3 tools/gpu/time_main_fusion.py 1000000 4 2>&1 | grep "main_fusion:" > $O/main_fusion_tiles_serial.log
cat $O/main_fusion_tiles_serial.log
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 --passes 0,4,5 $O/svx.json "f4l::,rocprim::,fillBuffer,copyBuffer" -- python3 tools/gpu/svx_only.py 10000000 3 > $O/svx_pmc.log 2>&1
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 --counters "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM" $O/svx_eval_extra.json "eval16_kernel,xch_eval_kernel" -- python3 tools/gpu/svx_only.py 10000000 3 > $O/svx_pmc_extra.log 2>&1
tail -2 $O/svx_pmc.log | cut -c1-600
