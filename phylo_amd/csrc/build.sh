#!/bin/bash
# Build libphylo_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
    -Wall -Wno-unused-function \
    phylo_hip.hip -o libphylo_hip.so -L/opt/rocm/lib -lrccl -lrt "$@"
echo "built $(pwd)/libphylo_hip.so"
