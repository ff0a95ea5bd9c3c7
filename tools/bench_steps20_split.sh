#!/bin/bash
# the driver's 20 steps as launch sets of different sizes: bash tools/bench_steps20_split.sh
for b in 20 10 5 4; do
  for s in 3 2; do
    timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --batch $b --streams $s --no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 800 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b streams $s', '%.3e' % d['value'], d['ms_per_step'])"
  done
done
