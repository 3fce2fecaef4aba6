#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on ONE GPU: two ranks share the card, the process group is gloo (C12381_BENCH_BACKEND),
# so the sharded legs run exactly the code the 8-GPU node runs except for the collective's transport.
# usage (GPU box): bash tools/bench_2rank_gloo.sh [extra bench args] > out.json
export C12381_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 2 --warmup 1 "$@"
