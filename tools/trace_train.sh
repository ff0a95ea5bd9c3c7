#!/bin/bash
# kernel stats of VI training steps: bash tools/trace_train.sh [train_probe args]   (through gpurun, from the repo root)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_train
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$REPO/tools/train_probe.py" --steps 10 "$@" > "$OUT/log.txt" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
grep "^\"pg_\|pk_rank_merge\|trampoline" "$f" | cut -d, -f1-4,6-7 | sed 's/rocprim::ROCPRIM_400200_NS::detail:://g' | cut -c1-130 | head -24
rm -rf "$OUT"
