python -m pytest tests -m gpu -q -x -k "icp or smoke" 2>&1 | tail -5
for w in 4 2; do for d in 0; do echo "waves=$w debug=$d"; F4L_ICP_PROF=1 F4L_ICP_PROF_WG=1 F4L_ICP_WAVES=$w F4L_ICP_DEBUG=$d python bench.py --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | grep -A1 "icp prof" | tail -4; done; done
for w in 4 2 1; do echo "waves=$w"; F4L_ICP_WAVES=$w python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_fitness'])"; done
