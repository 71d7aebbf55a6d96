#!/bin/bash
# round 6, call an: the headline shape at two and four waves per SIMD (256 / 128 registers) under the default scheduler
mkdir -p gpurun_out/r6an
TAIL=1 timeout -k 10 900 bash tools/gpu/lib_ab.sh "timeout -k 10 200 python bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 30 --warmup 5" icp_wide_wpe2 icp_wide_wpe4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end=' ')
    elif l.startswith('{'): d=json.loads(l); print('ms_per_step', d['ms_per_step'])
" | tee gpurun_out/r6an/icp_wide_wpe.log
