#!/bin/bash
mkdir -p gpurun_out/r6j
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j
one() { python3 bench.py --config $1 --cpu-seconds 0 --extras 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$2', '$1', d['ms_per_step'], 'ms', d['value'], 'Mpts/s fitness', d['config'].get('mean_fitness'))"; }
{
for rep in 1 2; do
for CFG in C3_10M_20k C2_1M_2k C4_50M_100k C1_50k_64; do
  one $CFG product
  F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prev.so one $CFG previous
  F4L_ICP_DEBUG=2048 one $CFG one_stage_switch
done; done
python3 tools/gpu/realistic_tile.py 2>&1 | tail -4
F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prev.so python3 tools/gpu/realistic_tile.py 2>&1 | tail -4
} > $O/two_stage_ab.log 2>&1
cat $O/two_stage_ab.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fine_matching.py tests/test_gpu_mirrors.py tests/test_gpu_randomised.py tests/test_gpu_fullsize.py -x -q -k "not C5" 2>&1 | tail -8 > $O/tests_icp.log
tail -6 $O/tests_icp.log
