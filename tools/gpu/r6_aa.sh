#!/bin/bash
# round 6, call aa: the bulk ICP shapes (point-to-point and point-to-plane at C4) under other instruction schedulers of the compiler
mkdir -p gpurun_out/r6aa
TAIL=3 timeout -k 10 1200 bash tools/gpu/lib_ab.sh "timeout -k 10 200 python tools/gpu/p2pl_phases.py" icp_sched_default icp_sched_itilp icp_sched_memclause icp_sched_minreg 2>&1 | tee gpurun_out/r6aa/icp_bulk_schedulers.log
