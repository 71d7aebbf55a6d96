#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/full_path_trace.sh <outdir> [n_points] -- every kernel of ONE pipeline.full_path call in launch
# order with its duration (rocprofv3 --kernel-trace), for the per-stage view of the glue around the partition and the patch loop.
OUT="${1:?usage: $0 <outdir> [n]}"; N="${2:-10000000}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/gpu/full_path_only.py $N 2 > $OUT/trace.log 2>&1
cp $OUT/t/*/*_kernel_trace.csv $OUT/full_path_kernel_trace.csv; rm -rf $OUT/t
python3 tools/gpu/full_path_trace_view.py $OUT/full_path_kernel_trace.csv
