python -m pytest tests -m gpu -q -x -k "icp" 2>&1 | tail -3
python tools/gpu/realistic_tile.py 2>&1 | grep "fixed iters" | tail -5
F4L_ICP_SERIAL_CLASSES=1 python tools/gpu/realistic_tile.py 2>&1 | grep "fixed iters" | tail -5
for e in 0 1; do echo -n "C3 serial=$e: "; env $( [ $e = 1 ] && echo F4L_ICP_SERIAL_CLASSES=1 ) python bench.py --config C3_10M_20k --steps 5 --warmup 2 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
