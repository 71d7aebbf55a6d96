#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/sv_trace.sh <outdir> [n] [res]   -- kernel stats + trace of tools/gpu/sv_only.py
OUT="${1:?usage: $0 <outdir> ...}"; N=${2:-1000000}; RES=${3:-1.386}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it: the root of the snapshot)}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sv -- python3 tools/gpu/sv_only.py $N $RES 4 > $OUT/sv.log 2>&1
cp $OUT/sv/*/*_kernel_stats.csv $OUT/sv_kernel_stats.csv
cp $OUT/sv/*/*_kernel_trace.csv $OUT/sv_kernel_trace.csv
rm -rf $OUT/sv
grep segmentation $OUT/sv.log
