#!/usr/bin/env python3
"""Per-wave instruction counts of the merge / twist kernels from one rocprofv3 --pmc run:
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d DIR -- python3 bench.py ...
   python tools/pmc_quick.py DIR"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if n.startswith("pk_rank_merge") or n.startswith("pk_twist") or n.startswith("void pp_"):
        acc[(n, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    c = {a: sum(b) / len(b) for a, b in v.items()}
    w = max(c.get("SQ_WAVES", 1), 1)
    print("%-28s workgroups %6d launches %4d  per wave: VALU %.0f  SALU %.0f  LDS %.0f" % (
        k[0], k[1], len(v["SQ_WAVES"]), c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w))
