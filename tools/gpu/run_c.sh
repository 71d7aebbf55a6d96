cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_knn -- python3 $GRAFT_REPO_ROOT/tools/gpu/time_knn.py > $GRAFT_REPO_ROOT/gpurun_out/prof_knn.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/prof_knn.log
head -14 gpurun_out/prof_knn/*/*_kernel_stats.csv | cut -c1-150
