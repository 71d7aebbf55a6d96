#!/bin/bash
mkdir -p gpurun_out/r6d
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d
python3 tools/gpu/svx_ab_r6.py 1000000 10000000 > $O/svx_ab.log 2>&1
grep -E "hash|equal" $O/svx_ab.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests.log
tail -6 $O/tests.log
python3 tools/gpu/time_main_fusion.py 1000000 8 > $O/main_fusion_tiles.log 2>&1
grep -E "main_fusion:|Current tile" $O/main_fusion_tiles.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_svx -- python3 tools/gpu/svx_only.py 10000000 3 > $O/svx_10M.log 2>&1
cp $O/stats_svx/*/*_kernel_stats.csv $O/svx_10M_kernel_stats.csv; rm -rf $O/stats_svx
head -8 $O/svx_10M_kernel_stats.csv | cut -c1-150
