#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/profile_r5.sh <outdir> -- the rocprofv3 --kernel-trace --stats summaries and un-profiled timings committed
# under profiles/r5_h_* (the state at the end of round 5): the commands of profile_r4.sh plus the exact device segmentation.
OUT="${1:?usage: $0 <outdir>}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it: the root of the snapshot)}"
run() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  cp $OUT/$name/*/*_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
run C4_50M_100k python3 bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 50 --warmup 5
run C3_10M_20k python3 bench.py --config C3_10M_20k --cpu-seconds 0 --extras 0 --steps 50 --warmup 5
run C2_1M_2k python3 bench.py --config C2_1M_2k --cpu-seconds 0 --extras 0 --steps 50 --warmup 5
run partition_10M python3 tools/gpu/svp_only.py 10000000 3
run knn_10M python3 tools/gpu/knn_only.py 10000000 knn
run full_path_1M python3 tools/gpu/full_path_only.py 1000000 10
bash tools/gpu/sv_exact_stats.sh $OUT/sv_exact 1000000 1.386 > $OUT/sv_exact_1M.log 2>&1
cp $OUT/sv_exact/sv_exact_kernel_stats.csv $OUT/sv_exact_1M_kernel_stats.csv; rm -rf $OUT/sv_exact
python3 tools/gpu/full_path_only.py 1000000 5 > $OUT/full_path_1M.log 2>&1
python3 tools/gpu/full_path_only.py 10000000 3 > $OUT/full_path_10M.log 2>&1
python3 tools/gpu/full_path_only.py 100000000 2 > $OUT/full_path_100M.log 2>&1
python3 tools/gpu/time_supervoxel.py > $OUT/time_supervoxel.log 2>&1
F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_sizes.py > $OUT/sv_exact_sizes.log 2>&1
F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_closures.py > $OUT/sv_exact_closures.log 2>&1
python3 tools/gpu/realistic_tile.py > $OUT/realistic_tile_1M.log 2>&1
python3 tools/gpu/time_fine_matching.py > $OUT/time_fine_matching.log 2>&1
python3 tools/gpu/time_all_ops.py C4_50M_100k > $OUT/all_ops_C4.log 2>&1
F4L_ICP_SERIAL_CLASSES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pc -- python3 bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 20 --warmup 3 > $OUT/C4_per_class.log 2>&1
cp $OUT/pc/*/*_kernel_stats.csv $OUT/C4_per_class_kernel_stats.csv; rm -rf $OUT/pc
ls $OUT
