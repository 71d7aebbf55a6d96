#!/bin/bash
# round 6, last call: the GPU suite and the smoke test on the tree as committed
mkdir -p gpurun_out/r6ap
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6ap/tests.log
timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | tee gpurun_out/r6ap/smoke.log
