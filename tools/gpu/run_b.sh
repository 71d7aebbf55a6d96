python -m pytest tests -m gpu -q -x -k "icp or smoke or full_size" 2>&1 | tail -3
b() { python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_fitness'])"; }
echo spread; b
echo dense; F4L_ICP_DEBUG=32 b
