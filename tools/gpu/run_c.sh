bash tools/gpu/profile_round.sh r1_f > gpurun_out/profile_r1_f.log 2>&1
tail -1 gpurun_out/prof_r1_f/bench.json.log | cut -c1-600
cat gpurun_out/prof_r1_f/traffic_raw.json
head -6 gpurun_out/prof_r1_f/stats/*/*_kernel_stats.csv | cut -c1-160
