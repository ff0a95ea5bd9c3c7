#!/bin/bash
# HBM traffic and VALU issue of the merge launches OUTSIDE the untrained primate.p regime, through the
# library: the "flat" workload (all-gap rows under JC69: ~63 % of the particles survive the first resampling, children spread
# over all earlier nodes) at DS1's shape, K = 4096, one sweep per launch set, lazy and eager nodes; and primate.p with
# trained parameters.  Three counter passes each (SQ, FETCH_SIZE, WRITE_SIZE).  tools/summarize_regimes.py reads the result.
#   tools/profile_regimes.sh r03
# synth1024_*: BASELINE config 5 at its per-GPU share (K = 8192 over 8 GPUs = 1024 particles, 208 GB node pool).
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$REPO/gpurun_out/prof_${TAG}_regimes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py"
COMMON="--no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0 --streams 1 --batch 1 --steps 4 --warmup 2"
run() { local rn=$1; shift; echo "== $rn"; "$@" > "$OUT/$rn.log" 2>&1 || echo "   (exit $?)"; }
python3 "$REPO/tests/probe_regimes.py" --epochs 40 --out "$OUT/regimes.json" --params-out "$OUT/trained_params.npz" > "$OUT/regime_probe.log" 2>&1
for cfg in "flat_lazy:--flat --dataset hohna_data_1 --n_particles 4096 --jcmodel true" \
           "flat_eager:--flat --dataset hohna_data_1 --n_particles 4096 --jcmodel true --eager" \
           "trained_lazy:--params $OUT/trained_params.npz" \
           "trained_eager:--params $OUT/trained_params.npz --eager" \
           "synth_lazy:--synthetic 128,50000 --n_particles 256" \
           "synth_eager:--synthetic 128,50000 --n_particles 256 --eager" \
           "synth1024_lazy:--synthetic 128,50000 --n_particles 1024" \
           "synth1024_eager:--synthetic 128,50000 --n_particles 1024 --eager"; do
  name=${cfg%%:*}; args=${cfg#*:}
  run ${name}_trace rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_trace" -- python3 "$B" $COMMON $args
  run ${name}_sq rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/${name}_sq" -- python3 "$B" $COMMON $args
  run ${name}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${name}_fetch" -- python3 "$B" $COMMON $args
  run ${name}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${name}_write" -- python3 "$B" $COMMON $args
done
echo "profiles in $OUT"
