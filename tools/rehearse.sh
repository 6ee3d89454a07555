#!/bin/bash
# Dev tool: the multi-rank bench flow on ONE GPU (gloo instead of RCCL, both ranks on device 0).  Inline exchange only:
# gloo's CUDA path stalls (and once hung) with the side-stream exchange, which says nothing about RCCL.
VC_BENCH_BACKEND=gloo VC_BENCH_ONE_GPU=1 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --db-size 2e8 --steps 10 --warmup 2 2>&1 | grep -E "bench rank|metric" | cut -c1-1400
