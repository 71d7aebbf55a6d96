#!/bin/bash
mkdir -p gpurun_out/r6k
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k
timeout 1500 python3 tools/gpu/fuzz_supervoxel_exact.py 160 1000 > $O/fuzz_svx.log 2>&1
tail -4 $O/fuzz_svx.log; grep -c ok $O/fuzz_svx.log; grep MISMATCH $O/fuzz_svx.log | head
timeout 600 python3 tools/gpu/fuzz_supervoxel_exact.py 6 5000 big > $O/fuzz_svx_big.log 2>&1
tail -3 $O/fuzz_svx_big.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
