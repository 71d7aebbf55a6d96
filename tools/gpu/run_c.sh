for rep in 1 2 3; do for v in old new; do
  cp tools/gpu/ab/$v.so fusion4landslide_amd/lib/libf4l_hip.so
  echo -n "$v: "; python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['extras']['fast_mode_f32']['value'])"
done; done
cp tools/gpu/ab/new.so fusion4landslide_amd/lib/libf4l_hip.so
python -m pytest tests -m gpu -q -x -k "icp or patch or nn_refine" 2>&1 | tail -3
