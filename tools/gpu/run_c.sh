F4L_ICP_PROF=1 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --extras 0 2>&1 | grep "icp prof\]" | tail -1
