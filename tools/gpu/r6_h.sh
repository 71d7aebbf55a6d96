#!/bin/bash
mkdir -p gpurun_out/r6h
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6h/tests.log
tail -4 gpurun_out/r6h/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 > gpurun_out/r6h/smoke.log; cat gpurun_out/r6h/smoke.log
bash tools/gpu/profile_r6.sh gpurun_out/r6h/prof > gpurun_out/r6h/profile.log 2>&1
tail -3 gpurun_out/r6h/profile.log
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gpu/slabs_two_ranks.py 1000000 2>&1 | grep "rank \|whole\|OK\|Error" | cut -c1-300 > gpurun_out/r6h/slabs_two_ranks.log; cat gpurun_out/r6h/slabs_two_ranks.log | tail -3
