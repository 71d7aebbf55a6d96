#!/bin/bash
mkdir -p gpurun_out/r6i
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i
one() { python3 bench.py --config C3_10M_20k --cpu-seconds 0 --extras 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], 'ms', d['value'], 'Mpts/s fitness', d['config'].get('mean_fitness'))"; }
{
one base
for SD in 12 16 24 32; do for DN in 2 1 0.5; do F4L_ICP_SUBDIV=$SD F4L_ICP_DENS=$DN one "subdiv=$SD dens=$DN"; done; done
for MC in 0.03 0.125 0.25; do F4L_ICP_MU_CELL=$MC one "mu_cell=$MC"; done
for SD in 16 24; do for MC in 0.125 0.25; do F4L_ICP_SUBDIV=$SD F4L_ICP_DENS=1 F4L_ICP_MU_CELL=$MC one "subdiv=$SD dens=1 mu_cell=$MC"; done; done
F4L_ICP_THROUGHPUT=1 one "throughput shape"
F4L_ICP_XSUB=4 one "xsub=4"
} > $O/c3_sweep.log 2>&1
cat $O/c3_sweep.log
ICP_PHASES_COUNTERS=1 F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prof.so python3 tools/gpu/icp_phases.py C3_10M_20k > $O/c3_phases_counters.log 2>&1
F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prof.so python3 tools/gpu/icp_phases.py C3_10M_20k > $O/c3_phases.log 2>&1
grep -E "class|icp prof" $O/c3_phases.log | cut -c1-420
grep -E "icp prof" $O/c3_phases_counters.log | cut -c1-420
