#!/bin/bash
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/gpu/svx_ab_r6.py 1000000 10000000 > gpurun_out/r6b/svx_ab.log 2>&1
tail -12 gpurun_out/r6b/svx_ab.log
python -m pytest tests/test_gpu_supervoxel_exact.py tests/test_gpu_supervoxel_parallel.py -x -q 2>&1 | tail -15 > gpurun_out/r6b/tests_sv.log
tail -5 gpurun_out/r6b/tests_sv.log
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_supervoxel_exact.py --deselect tests/test_gpu_supervoxel_parallel.py 2>&1 | tail -25 > gpurun_out/r6b/tests.log
tail -5 gpurun_out/r6b/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6b/stats_svx -- python3 tools/gpu/svx_only.py 10000000 3 > gpurun_out/r6b/svx_10M.log 2>&1
cp gpurun_out/r6b/stats_svx/*/*_kernel_stats.csv gpurun_out/r6b/svx_10M_kernel_stats.csv; rm -rf gpurun_out/r6b/stats_svx
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6b/stats_svx -- python3 tools/gpu/svx_only.py 1000000 3 > gpurun_out/r6b/svx_1M.log 2>&1
cp gpurun_out/r6b/stats_svx/*/*_kernel_stats.csv gpurun_out/r6b/svx_1M_kernel_stats.csv; rm -rf gpurun_out/r6b/stats_svx
grep f4l_supervoxel gpurun_out/r6b/svx_1*M.log
