#!/bin/bash
# streams x batch grid of the default bench (no parity / cpu baseline / vi step): bash tools/bench_grid.sh "3" "30 40 60" 120
for st in ${1:-2 3 4 6}; do for b in ${2:-10 20 30}; do
  steps=${3:-$((b*6))}
  timeout -k 10 120 python bench.py --steps $steps --warmup $b --streams $st --batch $b --no-cpu-baseline --no-parity --no-vi-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st batch $b steps $steps', '%.4g'%d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
