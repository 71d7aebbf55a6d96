#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/knn_stats.sh <outdir> [lib variant ...] -- mean duration of knn_lanes_kernel (rocprofv3 kernel stats) of
# tools/gpu/knn_only.py 10 M (f4l_knn: <false>) and tools/gpu/svp_only.py 10 M (the partition: <true>), product and variants.
OUT="${1:?}"; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for V in product "$@"; do
  if [ "$V" = product ]; then export F4L_LIB_PATH=""; else export F4L_LIB_PATH="$PWD/fusion4landslide_amd/lib/variants/lib_$V.so"; fi
  for T in "knn_only.py 10000000 knn" "svp_only.py 10000000 3"; do
    rm -rf $OUT/t; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/gpu/$T > $OUT/log.txt 2>&1
    python3 - "$V" "$T" $OUT/t <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[3] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "knn_lanes_kernel" in r["Name"]:
        print("%-10s %-28s %s: %.1f us mean of %s calls" % (sys.argv[1], sys.argv[2], r["Name"][10:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
  done
done
