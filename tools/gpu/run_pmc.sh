cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --extras 0 > gpurun_out/pmc/p$i.log 2>&1
  echo "set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for d in ("p1","p2","p3"):
    fs=glob.glob(f"gpurun_out/pmc/{d}/*/*_counter_collection.csv")
    if not fs: print(d,"none"); continue
    acc=collections.defaultdict(list); dur=[]
    for row in csv.DictReader(open(fs[0])):
        if "icp_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur.append(int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
    print(d, "kernel ns mean", sum(dur)/max(1,len(dur)))
    for k,v in sorted(acc.items()): print("  ",k,"mean=%.4g"%(sum(v)/len(v)))
    import json; json.dump({k: sum(v)/len(v) for k,v in acc.items()} | {"kernel_ns_mean": sum(dur)/max(1,len(dur))}, open(f"gpurun_out/pmc/{d}.json","w"), indent=1)
PY
