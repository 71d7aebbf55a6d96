#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu/profile_round.sh <tag>
# Produces under gpurun_out/prof_<tag>/ :
#   bench.json.log                      the un-profiled default `python3 bench.py` line (C4_50M_100k)
#   stats_<cfg>/ , stats_<cfg>.csv      rocprofv3 --kernel-trace --stats of `bench.py --config <cfg> --steps 50 --warmup 5`
#   fetch_<cfg>/ , write_<cfg>/         FETCH_SIZE and WRITE_SIZE of the same command in SEPARATE --pmc passes
#                                       (MI355X_MICROARCH.md: they do not fit one pass), 3 timed steps
#   stats_knn / fetch_knn / write_knn   the same three for tools/gpu/knn_only.py (f4l_knn and f4l_knn_normals, 1 M points, 13 launches each)
#   stats_sv                            kernel stats of tools/gpu/time_supervoxel.py (the device segmentation, 1 M points)
#   traffic_raw.json                    per-kernel means of the counters, in the counters' own unit
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench.json.log 2>$OUT/bench.err
for CFG in C4_50M_100k C3_10M_20k C2_1M_2k; do
  B="python3 bench.py --config $CFG --cpu-seconds 0 --extras 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$CFG -- $B --steps 50 --warmup 5 > $OUT/stats_$CFG.log 2>&1
  cp $OUT/stats_$CFG/*/*_kernel_stats.csv $OUT/stats_$CFG.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$CFG -- $B --steps 3 --warmup 1 > $OUT/fetch_$CFG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$CFG -- $B --steps 3 --warmup 1 > $OUT/write_$CFG.log 2>&1
done
K="python3 tools/gpu/knn_only.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_knn -- $K > $OUT/stats_knn.log 2>&1
cp $OUT/stats_knn/*/*_kernel_stats.csv $OUT/stats_knn.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_knn -- $K > $OUT/fetch_knn.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_knn -- $K > $OUT/write_knn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sv -- python3 tools/gpu/time_supervoxel.py > $OUT/stats_sv.log 2>&1
cp $OUT/stats_sv/*/*_kernel_stats.csv $OUT/stats_sv.csv
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for run in ("C4_50M_100k", "C3_10M_20k", "C2_1M_2k", "knn"):
    for name in ("fetch", "write"):
        fs = glob.glob(f"{out}/{name}_{run}/*/*_counter_collection.csv")
        acc = collections.defaultdict(list)
        if fs:
            for row in csv.DictReader(open(fs[0])):
                kn = row["Kernel_Name"]
                k = next((t for t in ("icp_kernel", "knn_lanes_kernel", "knn_listed_kernel", "knn_cells_kernel", "normals_kernel", "nn_refine_kernel", "sv_") if t in kn), None)
                if k:
                    if k == "sv_": k = kn.split("(")[0].split("::")[-1]
                    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            res.setdefault(run, {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(f"{out}/traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
for f in sorted(glob.glob(f"{out}/stats_*.csv")):
    print(f); print(open(f).read()[:1500])
PY
rm -rf $OUT/stats_*/ $OUT/fetch_*/ $OUT/write_*/
ls $OUT
