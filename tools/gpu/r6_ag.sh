#!/bin/bash
# round 6, call ag: icp.o (the shapes a tile and the dense patches run in) under other instruction schedulers
mkdir -p gpurun_out/r6ag
for c in C2_1M_2k C3_10M_20k; do
TAIL=1 timeout -k 10 900 bash tools/gpu/lib_ab.sh "timeout -k 10 200 python bench.py --config $c --cpu-seconds 0 --extras 0 --steps 40 --warmup 5" icp_m_minreg icp_m_memclause icp_m_itilp 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print('$c', l, end=' ')
    elif l.startswith('{'): d=json.loads(l); print('ms_per_step', d['ms_per_step'])
" | tee -a gpurun_out/r6ag/icp_main_schedulers.log
done
