#!/bin/bash
# bench.py as two ranks on ONE GPU (host-side collectives over shared memory): bash tools/bench_two_ranks.sh [bench args...]
# Rehearsal of the N > 1 path only: both ranks share the GPU, so the value is not a scaling figure.
D=$(mktemp -d)
export PHYLO_RDZV_DIR=$D MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29400 + RANDOM % 500)) PHYLO_COMM=hostshm WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 "$@" > $D/r1.out 2> $D/r1.err &
P1=$!
RANK=0 python bench.py --gpus 2 "$@" 2> $D/r0.err
wait $P1
rm -rf $D
