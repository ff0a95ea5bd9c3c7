#!/bin/bash
# kernel stats of one context's launch sets (the driver's --steps 20 regime): bash tools/trace_1ctx.sh  (through gpurun, from the repo root)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_1ctx
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PHYLO_BENCH_NO_TSWEEP=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 40 --warmup 4 --streams 1 --no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0 > "$OUT/log.txt" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 "$f" | cut -c1-110 | head -12
cp "$f" "$REPO/gpurun_out/trace_1ctx_stats.csv"
t=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for name in ('pk_sweep_prologue', 'pk_materialize_adopted_grouped', 'pk_rank_book_packed<8>', 'pp_resample_scan'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name'] and int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', 0)) > 0]
    print(name, len(d), ' '.join('%.1f' % x for x in d[-33:]))
PY
rm -rf "$OUT"
