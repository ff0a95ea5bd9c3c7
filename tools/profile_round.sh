#!/bin/bash
# Collect a round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 [quick]
# kernel trace + stats of the default bench command, of the one-context form whose merge launches bench.py prices, and of
# single sweeps (launches per rank event and the one-launch form); then FOUR separate counter passes (SQ counters, the fp64 /
# integer instruction classes, FETCH_SIZE, WRITE_SIZE: TCC has 4 slots, MI355X_MICROARCH.md) of the one-context form, which also issues single sweeps,
# so both launch shapes (40 960 and 2 048 particles) are priced.  The program itself follows `--` (no env / bash hop).
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py"
COMMON="--no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0"
run() { local rn=$1; shift; echo "== $rn"; "$@" > "$OUT/$rn.log" 2>&1 || echo "   (exit $?)"; }
run trace_default rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 "$B" --steps 40 --warmup 4 $COMMON
run trace_1ctx    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1ctx" -- python3 "$B" --steps 40 --warmup 4 --streams 1 $COMMON
run trace_1stream rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -- python3 "$B" --steps 40 --warmup 4 --streams 1 --batch 1 $COMMON
run trace_onelaunch rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_onelaunch" -- python3 "$B" --steps 40 --warmup 4 --streams 1 --batch 1 --one-launch $COMMON
run pmc_sq    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$B" --steps 20 --warmup 20 --streams 1 $COMMON
run pmc_f64   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d "$OUT/pmc_f64" -- python3 "$B" --steps 20 --warmup 20 --streams 1 $COMMON
run pmc_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$B" --steps 20 --warmup 20 --streams 1 $COMMON
run pmc_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$B" --steps 20 --warmup 20 --streams 1 $COMMON
if [ "${2:-}" != "quick" ]; then
run trace_twist rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_twist" -- python3 "$B" --twisting --M 1 --steps 6 --warmup 1 --streams 1 $COMMON
run pmc_twist   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_twist" -- python3 "$B" --twisting --M 1 --steps 3 --warmup 1 --streams 1 $COMMON
run pmc_twist_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_twist_fetch" -- python3 "$B" --twisting --M 1 --steps 3 --warmup 1 --streams 1 $COMMON
run pmc_twist_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_twist_write" -- python3 "$B" --twisting --M 1 --steps 3 --warmup 1 --streams 1 $COMMON
run pmc_twist_f64 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --output-format csv -d "$OUT/pmc_twist_f64" -- python3 "$B" --twisting --M 1 --steps 3 --warmup 1 --streams 1 $COMMON
run trace_ds1   rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ds1" -- python3 "$B" --dataset hohna_data_1 --n_particles 4096 --steps 6 --warmup 1 --streams 1 $COMMON
run trace_train rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -- python3 "$REPO/tools/train_probe.py" --steps 10
run trace_nested rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_nested" -- python3 "$REPO/tools/train_probe.py" --steps 10 --nested --M 1
fi
PROFILES_OUT=$REPO/gpurun_out/profiles_$TAG python3 "$REPO/tools/summarize_profiles.py" "$TAG" > "$OUT/summary.log" 2>&1
# the raw traces are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf "$OUT"/trace_* "$OUT"/pmc_*/ 2>/dev/null
echo "summaries in $REPO/gpurun_out/profiles_$TAG"
