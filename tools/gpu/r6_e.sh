#!/bin/bash
mkdir -p gpurun_out/r6e
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e
for L in 1 0; do
  for CFG in C4_50M_100k C2_1M_2k C3_10M_20k; do
    F4L_ICP_LAZY=$L python3 bench.py --config $CFG --cpu-seconds 0 --extras 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lazy=$L', '$CFG', d['ms_per_step'], 'ms', d['value'], 'Mpts/s fitness', d['config'].get('mean_fitness'))"
  done
done > $O/lazy_ab.log 2>&1
cat $O/lazy_ab.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fine_matching.py tests/test_gpu_mirrors.py tests/test_gpu_randomised.py tests/test_gpu_fullsize.py -x -q -k "not C5" 2>&1 | tail -30 > $O/tests_icp.log
tail -30 $O/tests_icp.log
