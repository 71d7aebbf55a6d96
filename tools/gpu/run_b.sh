bash tools/gpu/profile_round.sh r1_d > gpurun_out/profile_r1_d.log 2>&1
tail -1 gpurun_out/prof_r1_d/bench.json.log
cat gpurun_out/prof_r1_d/traffic_raw.json
head -4 gpurun_out/prof_r1_d/stats/*/*_kernel_stats.csv | cut -c1-150
