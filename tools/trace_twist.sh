#!/bin/bash
# kernel stats of twisted sweeps: bash tools/trace_twist.sh [M]   (through gpurun, from the repo root)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_twist
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --twisting --M ${1:-1} --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0 > "$OUT/log.txt" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4,6-7 "$f" | cut -c1-150 | head -12
rm -rf "$OUT"
