#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/sv_round.sh <outdir> [pmc]  -- the partition's parity tests, its time at 10 M points, its
# kernel stats, and (with `pmc`) the FETCH_SIZE / WRITE_SIZE passes of the same command: one iteration of the partition work of round 5.
OUT="${1:?usage: $0 <outdir> [pmc]}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_gpu_supervoxel_parallel.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/tests.log
python3 tools/gpu/svp_only.py 10000000 4 2>&1 | grep -v amdgpu.ids | tee $OUT/svp_10M.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/gpu/svp_only.py 10000000 3 > $OUT/kt.log 2>&1
cp $OUT/kt/*/*_kernel_stats.csv $OUT/partition_10M_kernel_stats.csv
python3 tools/gpu/sv_trace_view.py $OUT/kt/*/*_kernel_trace.csv 12 > $OUT/timeline_10M.log 2>&1; rm -rf $OUT/kt
if [ "$2" = "pmc" ]; then
  python3 tools/gpu/pmc_passes.py --counters "FETCH_SIZE;WRITE_SIZE" --sum-all --calls 3 $OUT/svp_bytes.json "f4l::,rocprim::,fillBuffer" -- python3 tools/gpu/svp_only.py 10000000 3 > $OUT/pmc.log 2>&1
fi
ls $OUT
