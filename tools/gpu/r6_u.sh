#!/bin/bash
# round 6, call u: XCD-contiguous loops in the parallel variant's gathering kernels (rows, sweep, min_metric, labels_init) -- tests, A/B, per-kernel
mkdir -p gpurun_out/r6u
timeout -k 10 400 python -m pytest tests/test_gpu_supervoxel_parallel.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6u/tests.log
grep -q "passed" gpurun_out/r6u/tests.log && ! grep -q "failed\|error" gpurun_out/r6u/tests.log || { echo "tests not green: stopping"; exit 1; }
TAIL=2 timeout -k 10 600 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svp_only.py 10000000 3" r6t_final > gpurun_out/r6u/svp_ab_10M.log 2>&1
cat gpurun_out/r6u/svp_ab_10M.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6u/prof10 -- python3 $R/tools/gpu/svp_only.py 10000000 3 > $R/gpurun_out/r6u/svp_10M_prof.log 2>&1
cd $R; f=$(find gpurun_out/r6u/prof10 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r6u/svp_10M_kernel_stats.csv && head -10 gpurun_out/r6u/svp_10M_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r6u/prof10
timeout -k 10 900 python3 tools/gpu/pmc_passes.py --sum-all --calls 3 gpurun_out/r6u/svp.json "f4l::,rocprim::ROCPRIM_400200,fillBuffer" -- python3 tools/gpu/svp_only.py 10000000 3 > gpurun_out/r6u/pmc.log 2>&1
