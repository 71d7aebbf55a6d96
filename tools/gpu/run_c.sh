python bench.py --steps 10 --warmup 3 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['cpu_baseline'], d['cpu_baseline_supervoxel'])"
