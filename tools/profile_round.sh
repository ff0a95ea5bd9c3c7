#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command and of the single-stream form, and the two PMC passes
#   (FETCH_SIZE, WRITE_SIZE -- separate runs, TCC has 4 slots) that price the merge kernel's HBM traffic.
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 "$REPO/bench.py" --steps 30 --warmup 3 --no-cpu-baseline > "$OUT/trace_default.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -- python3 "$REPO/bench.py" --steps 30 --warmup 3 --streams 1 --batch 1 --no-cpu-baseline > "$OUT/trace_1stream.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1ctx" -- python3 "$REPO/bench.py" --steps 30 --warmup 3 --streams 1 --no-cpu-baseline > "$OUT/trace_1ctx.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 10 --warmup 10 --streams 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 10 --warmup 10 --streams 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_twist" -- python3 "$REPO/bench.py" --twisting --M 1 --steps 6 --warmup 1 --streams 1 --no-cpu-baseline > "$OUT/trace_twist.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ds1" -- python3 "$REPO/bench.py" --dataset hohna_data_1 --n_particles 4096 --steps 6 --warmup 1 --streams 1 --no-cpu-baseline > "$OUT/trace_ds1.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -- python3 "$REPO/tools/train_probe.py" --steps 10 > "$OUT/trace_train.log" 2>&1
echo "profiles in $OUT"
