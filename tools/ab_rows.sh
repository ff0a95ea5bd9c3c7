#!/bin/bash
# A/B of the adopted nodes' chain: one launch (default) against a launch per rank event: bash tools/ab_rows.sh
for i in 1 2 3; do
  for v in all chain; do
    if [ $v = chain ]; then export PHYLO_GRAD_ROWS_CHAIN=1; else unset PHYLO_GRAD_ROWS_CHAIN; fi
    timeout -k 10 120 python tools/train_probe.py --steps 40 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v primate bw %.4f step %.4f min %.4f' % (d['backward_ms'], d['step_wall_ms'], d['step_wall_ms_min']))"
    timeout -k 10 120 python tools/train_probe.py --dataset hohna_data_1 --K 4096 --steps 12 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v ds1 bw %.4f step %.4f min %.4f' % (d['backward_ms'], d['step_wall_ms'], d['step_wall_ms_min']))"
  done
done
