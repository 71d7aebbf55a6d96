bash tools/gpu/profile_round.sh r1_g > gpurun_out/profile_r1_g.log 2>&1
tail -1 gpurun_out/prof_r1_g/bench.json.log | cut -c1-300
cat gpurun_out/prof_r1_g/traffic_raw.json
head -4 gpurun_out/prof_r1_g/stats/*/*_kernel_stats.csv | cut -c1-160
python tools/gpu/realistic_tile.py > gpurun_out/realistic_tile.log 2>&1; tail -22 gpurun_out/realistic_tile.log
python tools/gpu/scale_p.py > gpurun_out/scale_p.log 2>&1; tail -8 gpurun_out/scale_p.log
