#!/bin/bash
# the driver's command a few times (no CPU baseline, no VI step): bash tools/bench_steps20.sh [label] [runs]
for i in $(seq ${2:-2}); do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 800 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-run}', '%.3e' % d['value'], d['ms_per_step'])"
done
