cp tools/gpu/ab/new.so fusion4landslide_amd/lib/libf4l_hip.so
python -m pytest tests -m gpu -q -x -k "icp or nn_refine or patch or full_size" 2>&1 | tail -5
for v in old new; do
  cp tools/gpu/ab/$v.so fusion4landslide_amd/lib/libf4l_hip.so
  echo -n "$v C3: "; python bench.py --config C3_10M_20k --steps 5 --warmup 2 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo -n "$v C2: "; python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
