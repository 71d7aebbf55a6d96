#!/bin/bash
# round 6, call r: per-kernel times of f4l_supervoxel at 10 M points after the XCD-contiguous loops
mkdir -p gpurun_out/r6r
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6r/prof10 -- python3 $R/tools/gpu/svx_only.py 10000000 3 > $R/gpurun_out/r6r/svx_10M_prof.log 2>&1
cd $R; f=$(find gpurun_out/r6r/prof10 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r6r/svx_10M_kernel_stats.csv && head -14 gpurun_out/r6r/svx_10M_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r6r/prof10
