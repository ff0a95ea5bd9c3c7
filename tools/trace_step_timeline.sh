#!/bin/bash
# timeline of the last VI training step (every kernel: start, end, duration, queue): bash tools/trace_step_timeline.sh [vi_step_trace args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_tl
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/tools/vi_step_trace.py" "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('pk_sweep_prologue')][-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    n = r['Kernel_Name']
    if n.startswith('pk_rank_book') or n.startswith('pp_resample') or (n.startswith('pk_rank_merge') and False):
        continue
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    n = re.sub(r'rocprim::ROCPRIM_400200_NS::', '', n)
    n = re.sub(r'void detail::trampoline_kernel<detail::wrapped_', 'rp:', n)
    print("%8.1f %8.1f %7.1f q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], n[:50]))
PY
rm -rf "$OUT"
