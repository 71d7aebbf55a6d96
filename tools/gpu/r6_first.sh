#!/bin/bash
# round 6, first call: GPU suite + smoke + default bench line, then the default partition's kernel stats and counter passes
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6a/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 > gpurun_out/r6a/smoke.log
python bench.py 2> gpurun_out/r6a/bench.err > gpurun_out/r6a/bench.json.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6a/stats_svx -- python3 tools/gpu/svx_only.py 10000000 3 > gpurun_out/r6a/svx_10M.log 2>&1
cp gpurun_out/r6a/stats_svx/*/*_kernel_stats.csv gpurun_out/r6a/svx_10M_kernel_stats.csv; rm -rf gpurun_out/r6a/stats_svx
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 gpurun_out/r6a/svx.json "f4l::,rocprim::,fillBuffer,copyBuffer" -- python3 tools/gpu/svx_only.py 10000000 3 > gpurun_out/r6a/svx_pmc.log 2>&1
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 --counters "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE;SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" gpurun_out/r6a/svx_eval_extra.json "eval_kernel,xch_eval_kernel" -- python3 tools/gpu/svx_only.py 10000000 3 > gpurun_out/r6a/svx_pmc_extra.log 2>&1
timeout 900 python3 tools/gpu/svx_100M_vs_host.py 100000000 > gpurun_out/r6a/svx_100M_vs_host.log 2>&1
tail -3 gpurun_out/r6a/tests.log gpurun_out/r6a/smoke.log gpurun_out/r6a/svx_100M_vs_host.log
