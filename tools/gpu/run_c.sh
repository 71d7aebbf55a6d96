python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['extras']['fast_mode_f32'])"
