#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/sv_ab.sh <outdir> [other-lib.so]  -- the supervoxel partition of 10 M points under the current
# build (and, given another build of the library, under that one): kernel stats of tools/gpu/svp_only.py 10000000 3.
OUT="${1:?usage: $0 <outdir> [lib.so]}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}"
run() {  # name
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$1" -- python3 tools/gpu/svp_only.py 10000000 3 > "$OUT/$1.log" 2>&1
  cp "$OUT/$1"/*/*_kernel_stats.csv "$OUT/${1}_kernel_stats.csv"; rm -rf "$OUT/$1"
  python3 - "$OUT/${1}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(sys.argv[1], "total per call %.2f ms" % (tot / 3e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:9]:
    print("  %-60s %4s calls %8.3f ms per partition" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs']) / 3e6))
PY
}
run new
if [ -n "$2" ]; then F4L_LIB_PATH="$2" run old; fi
