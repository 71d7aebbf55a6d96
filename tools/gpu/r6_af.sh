#!/bin/bash
mkdir -p gpurun_out/r6af
timeout -k 10 900 python3 bench.py > gpurun_out/r6af/bench_C4.json.log 2> gpurun_out/r6af/bench_C4.err; tail -c 600 gpurun_out/r6af/bench_C4.json.log
