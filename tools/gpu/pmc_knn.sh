#!/bin/bash
# SQ counters of the kNN kernels (python3 tools/gpu/time_knn.py), three --pmc passes; prints per-launch means of knn_lanes_kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/pmc_knn; mkdir -p gpurun_out/pmc_knn
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_knn/p$i -- python3 tools/gpu/time_knn.py > gpurun_out/pmc_knn/p$i.log 2>&1
  echo "set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("p1","p2","p3"):
    fs=glob.glob(f"gpurun_out/pmc_knn/{d}/*/*_counter_collection.csv")
    if not fs: print(d,"none"); continue
    acc=collections.defaultdict(list); dur=[]
    for row in csv.DictReader(open(fs[0])):
        if "knn_lanes_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur.append(int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
    print(d, "kernel ns mean", sum(dur)/max(1,len(dur)), "launches", len(dur)//max(1,len(acc)))
    for k,v in sorted(acc.items()):
        print("  ",k,"mean=%.4g"%(sum(v)/len(v))); out[k]=sum(v)/len(v)
json.dump(out, open("gpurun_out/pmc_knn/knn_lanes_sq_counters.json","w"), indent=1)
PY
rm -rf gpurun_out/pmc_knn/p1 gpurun_out/pmc_knn/p2 gpurun_out/pmc_knn/p3
