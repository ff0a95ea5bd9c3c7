#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_regimes/ (tools/profile_regimes.sh) -> profiles/<tag>_regimes.md + profiles/<tag>_regimes.json.
usage: python tools/summarize_regimes.py r02"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_%s_regimes" % tag)
dst = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")
N_SIMD, CLK, HBM = 1024, 2.4e9, 8.0e12
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


lines = ["# Sweeps outside the untrained primate.p regime, round %s" % tag, "",
         "`tests/probe_regimes.py` (library calls, hipEvent median of 24 single sweeps, each configuration bit-exact vs the C oracle on seed 0)",
         "and `tools/profile_regimes.sh` (rocprofv3, three counter passes per configuration of `bench.py --streams 1 --batch 1 ...`).", ""]
rj = os.path.join(src, "regimes.json")
out_json = {"round": tag}
if os.path.exists(rj):
    r = json.load(open(rj))
    out_json["regime_probe"] = r
    lines += ["## survivors, materialised nodes and single-sweep time", "",
              "Training (%d epochs of Adam through the device reverse pass, minibatch log Z-hat %.1f -> %.1f) does NOT lift the weight"
              % (r['epochs'], r['elbo_first'], r['elbo_last']),
              "degeneracy: on real data a handful of particles survive every resampling whatever the parameters (log-weights spread over",
              "hundreds of nats, SURVEY F6), so lazy nodes write ~100 of 22 528 nodes per sweep.  The flat workload is the opposite extreme.", "",
              "| workload | form | t_sweep ms | units/s | nodes materialised / sweep | distinct ancestors per rank event |", "|---|---|---|---|---|---|"]
    for c in r['configs']:
        lines.append("| %s | %s | %.4f | %.3e | %d of %d | %s |" % (c['parameters'], c['form'], c['t_sweep_ms'], c['units_per_s'],
                                                                    c['nodes_materialised_per_sweep'], c['K'] * (c['N'] - 2),
                                                                    " ".join(str(x) for x in c['distinct_ancestors_per_rank_event'])))
    lines.append("")
rows = []
for name, what, S in (("flat_lazy", "flat 27 x 1949, K = 4096, JC69, lazy nodes", 1949), ("flat_eager", "flat 27 x 1949, K = 4096, JC69, eager nodes", 1949),
                      ("trained_lazy", "primate.p trained parameters, K = 2048, lazy nodes", 898),
                      ("trained_eager", "primate.p trained parameters, K = 2048, eager nodes", 898),
                      ("synth_lazy", "synthetic 128 x 50 000, K = 256, lazy nodes", 50000),
                      ("synth_eager", "synthetic 128 x 50 000, K = 256, eager nodes", 50000),
                      ("synth1024_lazy", "synthetic 128 x 50 000, K = 1024 (BASELINE config 5's per-GPU share), lazy nodes", 50000),
                      ("synth1024_eager", "synthetic 128 x 50 000, K = 1024 (BASELINE config 5's per-GPU share), eager nodes", 50000)):
    ntiles = (S + 2047) // 2048                        # contract v5: one merge wave per (particle, tile of 2048 sites)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, legacy in (("sq", "trace_sq"), ("fetch", "trace_sq_fetch"), ("write", "trace_sq_fetch_write")):
        f = one("%s_%s/*/*_counter_collection.csv" % (name, d)) or one("%s_%s/*/*_counter_collection.csv" % (name, legacy))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k.startswith('pk_rank_merge') or k.startswith('pk_materialize_adopted'):
                acc[(k, int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    t = one("%s_trace/*/*_kernel_trace.csv" % name)
    if t:
        for r in csv.DictReader(open(t)):
            k = r['Kernel_Name'].split('(')[0]
            dur[(k, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    log = os.path.join(src, name + "_trace.log")
    bench = None
    if os.path.exists(log):
        js = [l for l in open(log).read().splitlines() if l.startswith('{"metric"')]
        if js:
            bench = json.loads(js[-1])
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        fz, wz = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
        hbm = (2.0 * fz + wz) * 1024.0 if fz is not None and wz is not None else None
        us = sum(dur[key]) / len(dur[key]) if dur.get(key) else None
        particles = key[1] // ntiles if key[0].startswith('pk_rank_merge') else None
        alg = 96.0 * particles * S if particles else None
        rows.append({"config": what, "kernel": key[0], "workgroups": key[1], "launches": max(len(v) for v in acc[key].values()),
                     "avg_us": us, "fetch_kb": fz, "write_kb": wz, "hbm_bytes": hbm, "alg_bytes": alg,
                     "hbm_frac": hbm / (us * 1e-6) / HBM if hbm and us else None,
                     "valu_frac": c['SQ_INSTS_VALU'] * 4 / (N_SIMD * CLK * us * 1e-6) if c.get('SQ_INSTS_VALU') and us else None,
                     "bench_t_sweep_ms": bench.get('t_sweep_ms') if bench else None, "bench_value": bench.get('value') if bench else None})
if rows:
    out_json["merge_launches"] = rows
    lines += ["## merge / materialise launches by regime (counters per launch; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024)", "",
              "Durations are those of the un-countered kernel-trace pass of the same command.  A grid of the large-node kernels is",
              "particles x site tiles, so `workgroups` is not always the particle count.  VALU issue frac: every VALU instruction priced at",
              "4 cycles (an upper bound of the issue time; the primate.p table in the round summary splits fp64 from the rest).", "",
              "| configuration | kernel | workgroups | launches | avg us | FETCH KB | WRITE KB | HBM MB | algorithmic MB | HBM frac of 8 TB/s | VALU issue frac |",
              "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        f = lambda x, fmt: (fmt % x) if x is not None else "-"
        lines.append("| %s | %s | %d | %d | %s | %s | %s | %s | %s | %s | %s |" % (
            r['config'], r['kernel'], r['workgroups'], r['launches'], f(r['avg_us'], "%.2f"), f(r['fetch_kb'], "%.0f"), f(r['write_kb'], "%.0f"),
            f(r['hbm_bytes'] / 1e6 if r['hbm_bytes'] else None, "%.1f"), f(r['alg_bytes'] / 1e6 if r['alg_bytes'] else None, "%.1f"),
            f(r['hbm_frac'], "%.3f"), f(r['valu_frac'], "%.3f")))
    lines.append("")
open(os.path.join(dst, "%s_regimes.md" % tag), "w").write("\n".join(lines) + "\n")
json.dump(out_json, open(os.path.join(dst, "%s_regimes.json" % tag), "w"), indent=1)
print("\n".join(lines))
