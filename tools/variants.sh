#!/bin/bash
# Development helper (run through gpurun): rebuild the library with each set of -D flags and print the bench's merge-launch time.
#   tools/variants.sh "-DPK_ROWS_VARIANT=0" "-DPK_ROWS_VARIANT=1" ...   (env VARIANT_ENV="PHYLO_MERGE_WPB=4" applies to all)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for flags in "$@"; do
  ./phylo_amd/csrc/build.sh $flags > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  for envs in ${VARIANT_ENVS:-none}; do
    if [ "$envs" = none ]; then e=""; else e="$envs"; fi
    out=$(env $e python bench.py --no-cpu-baseline --no-vi-step --no-parity ${BENCH_ARGS:-} 2>&1 | tail -1)
    echo "$flags [$e] $(echo "$out" | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('value %.4g  ms/step %.4f  t_sweep %.4f  merge %.2f us' % (j['value'], j['ms_per_step'], j['t_sweep_ms'], r['avg_launch_us']))" 2>&1)"
  done
done
./phylo_amd/csrc/build.sh > /dev/null 2>&1
