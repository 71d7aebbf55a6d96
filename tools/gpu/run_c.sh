nproc; cat /sys/fs/cgroup/cpu.max; python -c "import bench; print(bench.host_cores())"
python bench.py --steps 10 --warmup 3 --extras 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['cpu_baseline'])"
