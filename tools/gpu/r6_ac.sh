#!/bin/bash
# round 6, call ac: the headline shape's translation unit (icp_bulk.o) under a few more compiler options
mkdir -p gpurun_out/r6ac
TAIL=1 timeout -k 10 1200 bash tools/gpu/lib_ab.sh "timeout -k 10 200 python bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 30 --warmup 5" icp_b_o2 icp_b_memclause icp_b_nopostsched icp_b_nounroll 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end=' ')
    elif l.startswith('{'): d=json.loads(l); print('ms_per_step', d['ms_per_step'])
" | tee gpurun_out/r6ac/icp_bulk_options.log
