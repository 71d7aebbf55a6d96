python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for e in 1 0; do
  echo -n "C3 noclasses=$e: "; env $( [ $e = 1 ] && echo F4L_ICP_NOCLASSES=1 ) python bench.py --config C3_10M_20k --steps 3 --warmup 1 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | cut -c1-200
