#!/bin/bash
# GPU test suite, smoke() and one default bench.py line (run on the MI355X box through gpurun, from the repo root)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json.log
tail -5 gpurun_out/bench.err
# the N > 1 path of the full hot path with the HIP kernels: 2 ranks (one GPU each over RCCL where the box has them, else sharing
# the one GPU over gloo); launched from this shell, which has not touched the GPU
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gpu/slabs_two_ranks.py 1000000 2>&1 | grep "rank \|whole\|OK\|Error" | cut -c1-300 | tee gpurun_out/slabs_two_ranks.log
