#!/bin/bash
# round 6, call ao: the headline kernel's counter passes again (icp.hip gained a macro: the counters are hash-guarded to the source), then the bench line
mkdir -p gpurun_out/r6ao/pmc
timeout -k 10 900 python3 tools/gpu/pmc_passes.py gpurun_out/r6ao/pmc/icp.json icp_kernel -- python3 bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 3 --warmup 1 > gpurun_out/r6ao/pmc.log 2>&1
python3 tools/make_roofline_profiles.py gpurun_out/r6ao/pmc r6_tmp > /dev/null 2>&1
timeout -k 10 900 python3 bench.py > gpurun_out/r6ao/bench_C4.json.log 2> gpurun_out/r6ao/bench_C4.err; tail -c 300 gpurun_out/r6ao/bench_C4.json.log
