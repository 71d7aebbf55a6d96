#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/profile_r6.sh <outdir> -- the rocprofv3 --kernel-trace --stats summaries and un-profiled timings committed
# under profiles/r6_am_* (the state at the end of round 6): profile_r5.sh's commands with the default (identical) partition in the lead.
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
run sv_exact_10M python3 tools/gpu/svx_only.py 10000000 3
run sv_exact_1M python3 tools/gpu/svx_only.py 1000000 3
run partition_variant_10M python3 tools/gpu/svp_only.py 10000000 3
run knn_10M python3 tools/gpu/knn_only.py 10000000 knn
python3 tools/gpu/time_main_fusion.py 1000000 8 > $OUT/main_fusion_8_tiles.log 2>&1
F4L_ASYNC_IO=0 python3 tools/gpu/time_main_fusion.py 1000000 8 2>&1 | grep -E "main_fusion:|Current tile" > $OUT/main_fusion_8_tiles_serial_io.log
python3 tools/gpu/time_all_ops.py C4_50M_100k > $OUT/all_ops_C4.log 2>&1
F4L_SV_EXACT_DEBUG=1 python3 tools/gpu/svx_sizes.py > $OUT/sv_exact_sizes.log 2>&1
timeout 900 python3 tools/gpu/svx_100M_vs_host.py 100000000 > $OUT/svx_100M_vs_host.log 2>&1
bash tools/gpu/roofline_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
ls $OUT
# (the bench line last: bench.py reads the traffic of its roofline objects from profiles/*_counters.json, which are hash-guarded to the kernel
#  sources -- after a kernel change they are stale until tools/make_roofline_profiles.py has turned the passes above into new ones)
python3 tools/make_roofline_profiles.py $OUT/pmc r6_tmp > /dev/null 2>&1
python3 bench.py > $OUT/bench_C4.json.log 2>> $OUT/bench_C4.err
