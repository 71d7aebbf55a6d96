#!/bin/bash
# For the day a multi-GPU box is at hand (8-GPU runs are the driver's; this pool's boxes have one GPU): the RCCL path of
# bench.py on 2 GPUs -- patches LPT-sharded, every rank generating its own share, one all-gather of the per-patch results per
# step -- and `pipeline.full_path_slabs` (slab partition with halos, second epoch joined, per-patch loop where the patch lives) on the
# same N ranks (N = 8 on a full node: `bash tools/gpu/rccl_smoke.sh 8`; its CPU twin is tests/test_slabs_gloo.py at world 8).  Run from the repo root; nothing here re-execs a process that
# has touched the GPU (bench.py starts its ranks as a child torch.distributed.run before any HIP call).
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-2}
python3 bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0 --extras 0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 tools/gpu/slabs_two_ranks.py 1000000
