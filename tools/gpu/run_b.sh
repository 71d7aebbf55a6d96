python -m pytest tests -m gpu -q -x 2>&1 | tail -3
b() { python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_fitness'])"; }
echo newton; b
echo svd; F4L_ICP_DEBUG=128 b
