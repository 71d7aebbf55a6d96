#!/bin/bash
# round 6, call t: the exchange's node state as one 16-byte record (product) against four arrays (svx_r6s2; svx_r6q = before the first_keys change)
mkdir -p gpurun_out/r6t
timeout -k 10 300 python -m pytest tests/test_gpu_supervoxel_exact.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6t/tests.log
grep -q "passed" gpurun_out/r6t/tests.log && ! grep -q "failed\|error" gpurun_out/r6t/tests.log || { echo "tests not green: stopping"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mirrors.py -m gpu -x -q -k "supervoxel" 2>&1 | tail -4 | tee -a gpurun_out/r6t/tests.log
TAIL=2 timeout -k 10 600 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svx_only.py 10000000 3" svx_r6s2 svx_r6q > gpurun_out/r6t/svx_ab_10M.log 2>&1
cat gpurun_out/r6t/svx_ab_10M.log
TAIL=2 timeout -k 10 600 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svx_only.py 1000000 3" svx_r6s2 > gpurun_out/r6t/svx_ab_1M.log 2>&1
cat gpurun_out/r6t/svx_ab_1M.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6t/prof10 -- python3 $R/tools/gpu/svx_only.py 10000000 3 > $R/gpurun_out/r6t/svx_10M_prof.log 2>&1
cd $R; f=$(find gpurun_out/r6t/prof10 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r6t/svx_10M_kernel_stats.csv && head -8 gpurun_out/r6t/svx_10M_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r6t/prof10
