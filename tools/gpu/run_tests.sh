#!/bin/bash
# GPU test suite, smoke() and one default bench.py line (run on the MI355X box through gpurun, from the repo root)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/tests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json.log
tail -5 gpurun_out/bench.err
