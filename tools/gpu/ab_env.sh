#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/ab_env.sh "<configs>" "<env settings ...>"  -- the bench step under environment switches, e.g.
#   bash tools/gpu/ab_env.sh "C4_50M_100k C2_1M_2k" "F4L_ICP_DEBUG=0" "F4L_ICP_DEBUG=1536" "F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_x.so"
# One line per (config, setting): ms per step and M points/s of `bench.py --cpu-seconds 0 --extras 0`; two rounds, so that drift shows.
CFGS="$1"; shift
for rep in 1 2; do
  for CFG in $CFGS; do
    for SET in "$@"; do
      env $SET python3 bench.py --config $CFG --cpu-seconds 0 --extras 0 --steps ${STEPS:-30} --warmup 5 2>/tmp/ab_err.log |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CFG | $SET |', d['ms_per_step'], 'ms |', d['value'], 'Mpts/s')" || tail -3 /tmp/ab_err.log
    done
  done
done
