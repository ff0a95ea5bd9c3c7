#!/bin/bash
# Development helper (run through gpurun): for each set of -D flags rebuild the library and run the eager workloads; prints the
# storing merge's launch time.  tools/variants2.sh "" "-DPK_STORE_DEPTH=2" ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
C="--no-cpu-baseline --no-vi-step --no-parity --batch 1 --streams 1 --steps 6 --warmup 2 --eager"
for flags in "$@"; do
  ./phylo_amd/csrc/build.sh $flags > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  for w in "primate:" "flatDS1:--flat --dataset hohna_data_1 --n_particles 4096 --jcmodel true" "synth256:--synthetic 128,50000 --n_particles 256"; do
    name=${w%%:*}; args=${w#*:}
    out=$(python bench.py $C $args 2>&1 | tail -1)
    echo "[$flags] $name $(echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('t_sweep %.4f  merge %.2f us  alg %.0f GB/s' % (j['t_sweep_ms'], r['avg_launch_us'], r['alg_equiv_GBps']))" 2>&1)"
  done
done
./phylo_amd/csrc/build.sh > /dev/null 2>&1
