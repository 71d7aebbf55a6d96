#!/bin/bash
mkdir -p gpurun_out/r6am
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6am/tests.log
tail -4 gpurun_out/r6am/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 > gpurun_out/r6am/smoke.log; cat gpurun_out/r6am/smoke.log
timeout -k 10 2400 bash tools/gpu/profile_r6.sh gpurun_out/r6am/prof > gpurun_out/r6am/profile.log 2>&1
tail -3 gpurun_out/r6am/profile.log
