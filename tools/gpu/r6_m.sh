#!/bin/bash
mkdir -p gpurun_out/r6m
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m
TAIL=3 bash tools/gpu/lib_ab.sh "python3 tools/gpu/svx_only.py 10000000 3" svx_spacked > $O/svx_ab_10M.log 2>&1
TAIL=3 bash tools/gpu/lib_ab.sh "python3 tools/gpu/svx_only.py 1000000 3" svx_spacked > $O/svx_ab_1M.log 2>&1
grep -E "==|f4l_supervoxel" $O/svx_ab_10M.log $O/svx_ab_1M.log | cut -c1-150
python -m pytest tests/test_gpu_supervoxel_exact.py tests/test_gpu_supervoxel_parallel.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
timeout 900 python3 tools/gpu/fuzz_supervoxel_exact.py 60 3000 > $O/fuzz_svx.log 2>&1; tail -2 $O/fuzz_svx.log
