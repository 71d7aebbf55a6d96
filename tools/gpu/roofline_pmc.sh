#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu/roofline_pmc.sh <outdir>  -- the counter passes behind bench.py's roofline objects
# (rocprim::ROCPRIM_400200 = the library's rocPRIM; torch's own copy, ROCPRIM_400001, sorts the synthetic cloud in the harness and is left out since round 6)
OUT=$1; mkdir -p $OUT
python3 tools/gpu/pmc_passes.py $OUT/icp.json icp_kernel -- python3 bench.py --config C4_50M_100k --cpu-seconds 0 --extras 0 --steps 3 --warmup 1
python3 tools/gpu/pmc_passes.py --sum-all --calls 13 $OUT/knn.json "f4l::,rocprim::ROCPRIM_400200,fillBuffer" -- python3 tools/gpu/knn_only.py 10000000 knn
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 $OUT/svp.json "f4l::,rocprim::ROCPRIM_400200,fillBuffer" -- python3 tools/gpu/svp_only.py 10000000 3
python3 tools/gpu/pmc_passes.py --sum-all --calls 3 $OUT/svx.json "f4l::,rocprim::ROCPRIM_400200,fillBuffer,copyBuffer" -- python3 tools/gpu/svx_only.py 10000000 3
