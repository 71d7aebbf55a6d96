#!/bin/bash
# round 6, call s: min_metric / xch_first_keys with 16 lanes per point -- labels (tests), A/B against the build before, per-kernel times
mkdir -p gpurun_out/r6s
timeout -k 10 300 python -m pytest tests/test_gpu_supervoxel_exact.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6s/tests.log
grep -q "passed" gpurun_out/r6s/tests.log && ! grep -q "failed\|error" gpurun_out/r6s/tests.log || { echo "tests not green: stopping"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "supervoxel" 2>&1 | tail -4 | tee -a gpurun_out/r6s/tests.log
TAIL=3 timeout -k 10 400 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svx_only.py 10000000 3" svx_r6q > gpurun_out/r6s/svx_ab_10M.log 2>&1
cat gpurun_out/r6s/svx_ab_10M.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6s/prof10 -- python3 $R/tools/gpu/svx_only.py 10000000 3 > $R/gpurun_out/r6s/svx_10M_prof.log 2>&1
cd $R; f=$(find gpurun_out/r6s/prof10 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r6s/svx_10M_kernel_stats.csv && head -12 gpurun_out/r6s/svx_10M_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r6s/prof10
timeout -k 10 900 python3 tools/gpu/pmc_passes.py --sum-all --calls 3 gpurun_out/r6s/svx.json "f4l::,rocprim::ROCPRIM_400200,fillBuffer,copyBuffer" -- python3 tools/gpu/svx_only.py 10000000 3 > gpurun_out/r6s/pmc.log 2>&1
