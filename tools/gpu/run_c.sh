python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['extras'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])"
for c in C1_50k_64 C3_10M_20k C4_50M_100k; do echo -n "$c: "; python bench.py --config $c --steps 5 --warmup 2 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
python tools/gpu/realistic_tile.py 2>&1 | grep "fixed iters" | tail -4
