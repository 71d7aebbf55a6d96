#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/sv_pmc.sh <outdir>  -- counter passes of tools/gpu/sv_only.py, per-kernel means of the svg:: kernels
OUT="${1:?usage: $0 <outdir>}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it: the root of the snapshot)}"
i=0
# (a pass that asks for more counters of one block than the hardware has aborts inside rocprofv3 and never returns: two per pass, and a time limit)
SETS=${SV_PMC_SETS:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU|TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum|TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum|FETCH_SIZE|WRITE_SIZE"}
IFS='|' read -ra SETLIST <<< "$SETS"
for set in "${SETLIST[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/gpu/sv_only.py 1000000 1.386 2 > $OUT/p$i.log 2>&1
  echo "set $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{out}/p*/")):
    fs = glob.glob(f"{d}/*/*_counter_collection.csv")
    if not fs: continue
    rows = list(csv.DictReader(open(fs[0])))
    # per kernel name AND dispatch order within the last segmentation: key = name#occurrence
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "init_state_kernel" in r["Kernel_Name"]]
    seg = rows[starts[-1]:] if starts else rows
    occ = collections.Counter(); seen = {}
    for r in seg:
        nm = r["Kernel_Name"].split("(")[0].replace("f4l::svg::", "").replace("void ", "")
        did = r["Dispatch_Id"]
        if did not in seen:
            seen[did] = f"{nm}#{occ[nm]}"; occ[nm] += 1
        k = seen[did]
        res[k][r["Counter_Name"]] = float(r["Counter_Value"])
        res[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
json.dump(res, open(f"{out}/sv_counters.json", "w"), indent=1)
for k in ("build_kernel<true>#0", "build_kernel<true>#1", "build_kernel<false>#0", "sweep_kernel#0", "cand2_kernel#0", "apply_kernel#0", "min_metric_kernel#0"):
    if k in res: print(k, json.dumps(res[k]))
PY
rm -rf $OUT/p*/
