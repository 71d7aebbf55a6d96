python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu/profile_round.sh r1_c > gpurun_out/profile_r1_c.log 2>&1
tail -1 gpurun_out/prof_r1_c/bench.json.log
cat gpurun_out/prof_r1_c/traffic_raw.json
