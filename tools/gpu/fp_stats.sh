#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/fp_stats.sh <outdir> [n]  -- kernel stats of the whole path of one tile (pipeline.full_path)
OUT="${1:?}"; N=${2:-10000000}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp -- python3 tools/gpu/full_path_only.py $N 3 > $OUT/fp.log 2>&1
cp $OUT/fp/*/*_kernel_stats.csv $OUT/full_path_kernel_stats.csv
python3 - $OUT/fp/*/*_kernel_trace.csv > $OUT/full_path_timeline.log <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last full_path call: from the last bbox_kernel of f4l_knn that precedes a knn_lanes_kernel<true>
idx = [i for i, r in enumerate(rows) if "knn_lanes_kernel<true>" in r["Kernel_Name"]]
start = idx[-1]
while start > 0 and "f4l::bbox_kernel" not in rows[start]["Kernel_Name"]: start -= 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 60: print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {r['Kernel_Name'].split('(')[0][-70:]:70s} {d:9.1f} us")
P
rm -rf $OUT/fp; grep wall $OUT/fp.log | tail -1
