#!/bin/bash
# round 6, call ah: the two epochs of a tile partitioned on two streams -- entry tests, 8 tiles with and without
mkdir -p gpurun_out/r6ah
timeout -k 10 600 python -m pytest tests/test_fusion_entry.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r6ah/tests.log
for s in 2 1 2 1; do F4L_PARTITION_STREAMS=$s timeout -k 10 300 python3 tools/gpu/time_main_fusion.py 1000000 8 2>&1 | grep -E "main_fusion:|Current tile" | sed "s/^/streams=$s /" | tee -a gpurun_out/r6ah/main_fusion_streams.log | grep "main_fusion:"; done
