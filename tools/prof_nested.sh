#!/bin/bash
# tools/prof_nested.sh <tag> [train_probe args]: twisted-gradient tests, then the kernel stats of nested training steps (gpurun)
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-nested}; shift || true
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO" && timeout -k 10 300 python -m pytest tests/test_gpu_grad.py -x -q -s -k twisted > "$OUT/tests.log" 2>&1
grep -n "twisted K=\|passed\|failed\|Error\|assert" "$OUT/tests.log" | tail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tw" -- python3 "$REPO/tools/train_probe.py" --nested --steps 5 "$@" > "$OUT/probe.log" 2>&1
grep '^{' "$OUT/probe.log"
f=$(ls "$OUT"/tw/*/*kernel_stats.csv) && cp "$f" "$OUT/kernel_stats.csv" && head -14 "$f"
rm -rf "$OUT/tw"
