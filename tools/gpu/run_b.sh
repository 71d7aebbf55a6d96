bash tools/gpu/profile_round.sh r1_e > gpurun_out/profile_r1_e.log 2>&1
tail -1 gpurun_out/prof_r1_e/bench.json.log
cat gpurun_out/prof_r1_e/traffic_raw.json
head -4 gpurun_out/prof_r1_e/stats/*/*_kernel_stats.csv | cut -c1-150
bash tools/gpu/run_pmc.sh 2>&1 | tail -28
