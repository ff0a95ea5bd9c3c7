"""Timeline of the last training step in a rocprofv3 kernel trace: python tools/trace_tail.py <dir>"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('pg_copy_words')][-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    n = re.sub(r'rocprim::ROCPRIM_400200_NS::', '', r['Kernel_Name'])
    n = re.sub(r'void detail::trampoline_kernel<detail::wrapped_', 'rp:', n)
    print("%8.1f %8.1f %7.1f q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], n[:60]))
