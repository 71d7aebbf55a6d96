#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu/profile_round.sh <tag>
# Produces under gpurun_out/prof_<tag>/ : kernel-trace stats of `bench.py`, and the HBM traffic counters of
# icp_kernel (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes).
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json.log 2>$OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --extras 0 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --extras 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --extras 0 > $OUT/write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for name in ("fetch", "write"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    acc = collections.defaultdict(list)
    if fs:
        for row in csv.DictReader(open(fs[0])):
            k = "icp_kernel" if "icp_kernel" in row["Kernel_Name"] else ("kabsch_kernel" if "kabsch_kernel" in row["Kernel_Name"] else ("apply_transform" if "apply_transform" in row["Kernel_Name"] else None))
            if k: acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in acc.items():
        res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(f"{out}/traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
for f in glob.glob(f"{out}/stats/*/*_kernel_stats.csv"):
    print(open(f).read()[:3000])
PY
ls -R $OUT | head -40
