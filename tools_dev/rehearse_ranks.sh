#!/bin/bash
# usage (GPU box, from a shell that has not touched the GPU): tools_dev/rehearse_ranks.sh [ranks=2]
# bench.py's multi-rank control flow with all ranks sharing GPU 0 (HSD_COMM_DRYRUN=1: no RCCL communicator,
# signatures exchanged through gloo).  A rehearsal of the flow -- row shards, barriers, max over ranks, one
# JSON line -- not a measurement.  (Not a pytest: a test process that has initialised the GPU must not
# spawn further GPU programs on this pool.)
P=${1:-2}
HSD_COMM_DRYRUN=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus $P --steps 20 --warmup 3 --workload config2
