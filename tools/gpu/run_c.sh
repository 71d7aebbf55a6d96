python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['extras'])"
