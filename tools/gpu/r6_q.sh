#!/bin/bash
# round 6, call q: XCD-contiguous loops in the partition's gathering kernels (min_metric, xch_first_keys, xch_push, rootlists) and 64-entry
# workgroups in xch_eval_kernel -- labels (tests), A/B against the build before, per-kernel times.  Every step under its own timeout.
mkdir -p gpurun_out/r6q
timeout -k 10 300 python -m pytest tests/test_gpu_supervoxel_exact.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6q/tests.log
grep -q "passed" gpurun_out/r6q/tests.log && ! grep -q "failed\|error" gpurun_out/r6q/tests.log || { echo "tests not green: stopping"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "supervoxel or normals" 2>&1 | tail -4 | tee -a gpurun_out/r6q/tests.log
TAIL=3 timeout -k 10 300 bash tools/gpu/lib_ab.sh "timeout -k 10 120 python tools/gpu/svx_only.py 1000000 3" svx_r6n > gpurun_out/r6q/svx_ab_1M.log 2>&1
TAIL=3 timeout -k 10 400 bash tools/gpu/lib_ab.sh "timeout -k 10 150 python tools/gpu/svx_only.py 10000000 3" svx_r6n > gpurun_out/r6q/svx_ab_10M.log 2>&1
cat gpurun_out/r6q/svx_ab_1M.log gpurun_out/r6q/svx_ab_10M.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6q/prof10 -o run -- python3 $GRAFT_REPO_ROOT/tools/gpu/svx_only.py 10000000 3 > $GRAFT_REPO_ROOT/gpurun_out/r6q/svx_10M_prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/r6q/prof10/*/*kernel_stats.csv gpurun_out/r6q/prof10/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r6q/svx_10M_kernel_stats.csv && head -12 gpurun_out/r6q/svx_10M_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r6q/prof10
