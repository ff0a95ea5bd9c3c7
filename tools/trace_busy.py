"""GPU busy fraction and per-kernel share of a rocprofv3 kernel trace over its last `window_ms`: python tools/trace_busy.py <dir> [window_ms]"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in csv.DictReader(open(f))]
rows.sort()
t1 = max(e for _, e, _ in rows)
t0 = t1 - win
rows = [r for r in rows if r[1] > t0]
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = defaultdict(int)
for s, e, n in rows:
    tot[n] += e - max(s, t0)
print("window %.1f ms: some kernel running %.1f %% of the time; sum of kernel durations / window = %.2f" % (win / 1e6, 100.0 * busy / win, sum(tot.values()) / win))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-42s %.2f of the window" % (n, v / win))
