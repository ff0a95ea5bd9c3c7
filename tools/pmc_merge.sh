#!/bin/bash
# Quick counter passes of the merge kernel at the bench's launch shape (run through gpurun from the repo root):
#   tools/pmc_merge.sh TAG [extra bench args]
# SQ instruction mix (two passes: totals, then the fp64 / integer classes) of `bench.py --streams 1`; prints per-wave figures.
set -uo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift || true
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0 --steps 10 --warmup 10 --streams 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -- python3 "$REPO/bench.py" $COMMON "$@" > "$OUT/sq.log" 2>&1 || echo "sq pass: exit $?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES --output-format csv -d "$OUT/f64" -- python3 "$REPO/bench.py" $COMMON "$@" > "$OUT/f64.log" 2>&1 || echo "f64 pass: exit $?"
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for d in ("sq", "f64"):
    fs = glob.glob(out + "/" + d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("pk_rank_merge") or n.startswith("pk_twist") or n.startswith("void pp_"):
            acc[(n, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for t in glob.glob(out + "/" + d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(t)):
            n = r["Kernel_Name"].split("(")[0]
            dur[(n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items()):
        c = {a: sum(b) / len(b) for a, b in v.items()}
        w = max(c.get("SQ_WAVES", 1), 1)
        us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        print("%-26s wgs %6d n %4d avg %.2f us | per wave: " % (k[0], k[1], len(v["SQ_WAVES"]), us) +
              "  ".join("%s %.1f" % (a.replace("SQ_INSTS_", "").replace("SQ_", ""), c[a] / w) for a in sorted(c) if a != "SQ_WAVES"))
PY
rm -rf "$OUT/sq" "$OUT/f64"
