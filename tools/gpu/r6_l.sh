#!/bin/bash
mkdir -p gpurun_out/r6l
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l
timeout 1500 python3 tools/gpu/fuzz_supervoxel_exact.py 160 1000 > $O/fuzz_svx.log 2>&1
tail -3 $O/fuzz_svx.log; grep -c " ok" $O/fuzz_svx.log; grep -c REFUSED $O/fuzz_svx.log; grep MISMATCH $O/fuzz_svx.log | head
F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_only.py 1000000 1 2>&1 | grep "sv exact" > $O/pool_usage_1M.log
F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_only.py 10000000 1 2>&1 | grep "sv exact" > $O/pool_usage_10M.log
cat $O/pool_usage_10M.log | cut -c1-250
python -m pytest tests/test_gpu_supervoxel_exact.py -x -q 2>&1 | tail -3
