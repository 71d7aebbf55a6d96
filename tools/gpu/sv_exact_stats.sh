#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/sv_exact_stats.sh <outdir> [n] [res]  -- kernel stats of f4l_supervoxel (the reference's labels on the device)
OUT="${1:?}"; N=${2:-1000000}; RES=${3:-1.386}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
cat > /tmp/svx_run.py <<P
import sys, time, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from fusion4landslide_amd import engine, synthetic
n, res = $N, $RES
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
xyz = d["src"]
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); lab, K = engine.supervoxel(xyz, 30, res); torch.cuda.synchronize()
    print(f"f4l_supervoxel n={n} res={res}: {1e3 * (time.perf_counter() - t):.1f} ms K={K}", flush=True)
P
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 /tmp/svx_run.py > $OUT/run.log 2>&1
cp $OUT/kt/*/*_kernel_stats.csv $OUT/sv_exact_kernel_stats.csv; rm -rf $OUT/kt
grep f4l_supervoxel $OUT/run.log
python3 - $OUT/sv_exact_kernel_stats.csv <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {int(r['Calls'])//3:6d}  ms/call-of-3 {float(r['TotalDurationNs'])/3e6:8.3f}")
P
