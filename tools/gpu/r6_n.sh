#!/bin/bash
mkdir -p gpurun_out/r6n
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6n/tests.log
tail -4 gpurun_out/r6n/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 > gpurun_out/r6n/smoke.log; cat gpurun_out/r6n/smoke.log
bash tools/gpu/profile_r6.sh gpurun_out/r6n/prof > gpurun_out/r6n/profile.log 2>&1
tail -3 gpurun_out/r6n/profile.log
