#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/icp_trace.sh <outdir> <config> [env assignments...] -- kernel trace of a few bench steps
OUT="${1:?usage: $0 <outdir> ...}"; CFG=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it: the root of the snapshot)}"
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --config $CFG --cpu-seconds 0 --extras 0 --steps 4 --warmup 2 > $OUT/bench.log 2>&1
cp $OUT/t/*/*_kernel_trace.csv $OUT/kernel_trace.csv
rm -rf $OUT/t
tail -1 $OUT/bench.log | cut -c1-200
