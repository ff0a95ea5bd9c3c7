#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:
   profiles/<tag>_kernel_stats_<run>.csv   rocprofv3 --stats tables as produced
   profiles/<tag>_merge_pmc.json           per launch shape of the merge kernels: SQ counters, FETCH_SIZE / WRITE_SIZE and the
                                           HBM bytes derived from them (read by bench.py for its roofline block)
   profiles/<tag>_summary.md               per-kernel averages and the roofline arithmetic, with the commands
usage: python tools/summarize_profiles.py r02
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
S_SITES, WORKLOAD = 898, "primate.p"
N_SIMD, CLK, HBM = 1024, 2.4e9, 8.0e12
FP64_PEAK = 78.6e12          # vector fp64: 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz = half the guide's 157.3 TFLOP/s fp32 vector peak
FLOP_PER_UNIT = 60.0         # 2 x (16 mul + 12 add) + 4 mul per particle-site-likelihood


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return f[-1] if f else None          # gpurun merges runs into the same directory: take the newest


RUNS = [("default", "`python bench.py --steps 40 --warmup 4` (launch sets of 20 sweeps on the contexts in flight: the form of the timed region of the bench line)"),
        ("1ctx", "`--streams 1`: the same launch sets one at a time (the form whose merge launches bench.py prices)"),
        ("1stream", "`--streams 1 --batch 1`: one sweep at a time, launches per rank event"),
        ("onelaunch", "`--streams 1 --batch 1 --one-launch`: single sweeps (t_sweep section) in the one-launch form, phylo_persist.h"),
        ("twist", "`--twisting --M 1 --streams 1`"), ("ds1", "`--dataset hohna_data_1 --n_particles 4096 --streams 1`"),
        ("train", "`python tools/train_probe.py --steps 10` (sweep with the graph kept + reverse pass)"),
        ("nested", "`python tools/train_probe.py --steps 10 --nested --M 1` (twisted sweep with the graph kept + its reverse pass)")]
lines = ["# rocprofv3 summary, round %s" % tag, "",
         "Workload: primate.p N=12 S=898, GTR-init, K=2048 per sweep.  All bench commands carry",
         "`--no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0`.  Raw tables: `%s_kernel_stats_*.csv`; counters: `%s_merge_pmc.json`." % (tag, tag), ""]
for name, what in RUNS:
    st = one("trace_%s/*/*_kernel_stats.csv" % name)
    if not st:
        continue
    shutil.copy(st, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name)))
    lines += ["## kernel stats: %s" % name, "", what, "", "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(st)):
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %s |" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3,
                                                         float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
    tr = one("trace_%s/*/*_kernel_trace.csv" % name)
    if tr:                                                 # the merge kernels by launch shape
        shapes = collections.defaultdict(list)
        for r in csv.DictReader(open(tr)):
            if r['Kernel_Name'].startswith('pk_rank_merge') or r['Kernel_Name'].startswith('void pp_sweep'):
                shapes[(r['Kernel_Name'].split('(')[0], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append(
                    (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        if shapes:
            lines += ["", "by launch shape: " + "; ".join(
                "`%s` x %d workgroups: %d launches, avg %.2f us" % (k[0], k[1], len(v), sum(v) / len(v)) for k, v in sorted(shapes.items()))]
    logp = os.path.join(src, "trace_%s.log" % name)
    log = open(logp).read() if os.path.exists(logp) else ""
    js = [l for l in log.splitlines() if l.startswith('{"metric"')]
    if js:
        j = json.loads(js[-1])
        lines += ["", "bench line under the profiler: value %.4g %s, ms_per_step %.4f, t_sweep_ms %.4f (%s)"
                  % (j['value'], j['unit'], j['ms_per_step'], j.get('t_sweep_ms', float('nan')), j.get('t_sweep', {}).get('form', '')), ""]
    tj = [l for l in log.splitlines() if l.startswith('{"dataset"')]
    if tj:
        j = json.loads(tj[-1])
        lines += ["", "training step under the profiler: forward %.3f ms, reverse pass %.3f ms, step wall %.3f ms (K=%d, %d sites)"
                  % (j['forward_ms'], j['backward_ms'], j['step_wall_ms'], j['K'], j['sites']), ""]

# ---- counters of the merge kernels, by launch shape (particles per launch = workgroups)
acc = collections.defaultdict(lambda: collections.defaultdict(list))       # (kernel, particles) -> counter -> values
dur = collections.defaultdict(list)
for d in ("pmc_sq", "pmc_f64", "pmc_fetch", "pmc_write"):
    f = one("%s/*/*_counter_collection.csv" % d)
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0]
        if not name.startswith('pk_rank_merge'):
            continue
        key = (name, int(r['Grid_Size']) // int(r['Workgroup_Size']))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    t = one("%s/*/*_kernel_trace.csv" % d)
    if t and d == "pmc_sq":
        for r in csv.DictReader(open(t)):
            name = r['Kernel_Name'].split('(')[0]
            if name.startswith('pk_rank_merge'):
                dur[(name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
shapes_out = []
if acc:
    lines += ["## counters of the merge kernel, per launch (four separate rocprofv3 --pmc passes of the `1ctx` command)", "",
              "FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them; on gfx950 FETCH_SIZE reads half the bytes of a wide coalesced",
              "stream (MI355X_MICROARCH.md, HBM), so HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- an upper bound here, where much",
              "of the read side is 1-byte codes.  Work fraction = 60 flop x units / duration / 78.6 TFLOP/s (algorithmic flops, SURVEY 8d);",
              "executed fp64 = (2 FMA + ADD + MUL) x 64 lanes from the SQ_INSTS_VALU_*_F64 counters; VALU issue occupancy prices fp64",
              "instructions at 4 cycles and every other VALU instruction at 2 (a lower bound: shifts, compares, bit-field and DPP ops",
              "measure 4, tools/ubench) over 1024 SIMDs x 2.4 GHz x duration.  Durations in this table are those of the COUNTER pass",
              "(serialised dispatches); bench.py uses its own live launch time.", "",
              "| kernel | particles / launch | launches | VALU / wave | fp64 FMA / ADD / MUL per wave | INT32 / wave | SALU / wave | LDS / wave | wait share | FETCH KB | WRITE KB | HBM MB | algorithmic MB | avg us (pmc pass) | work frac (60 flop) | executed fp64 frac | VALU issue occupancy | HBM frac |",
              "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        n = max(len(v) for v in acc[key].values())
        fz, wz = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
        hbm = (2.0 * fz + wz) * 1024.0 if fz is not None and wz is not None else None
        alg = 96.0 * key[1] * S_SITES
        us = sum(dur[key]) / len(dur[key]) if dur.get(key) else None
        valu = c.get('SQ_INSTS_VALU')
        w = c.get('SQ_WAVES') or 1.0
        fma, add, mul = c.get('SQ_INSTS_VALU_FMA_F64'), c.get('SQ_INSTS_VALU_ADD_F64'), c.get('SQ_INSTS_VALU_MUL_F64')
        f64 = (fma + add + mul) if fma is not None and add is not None and mul is not None else None
        units = float(key[1]) * S_SITES
        wf = FLOP_PER_UNIT * units / (us * 1e-6) / FP64_PEAK if us else None
        ef = (2.0 * fma + add + mul) * 64.0 / (us * 1e-6) / FP64_PEAK if f64 is not None and us else None
        vf = (4.0 * f64 + 2.0 * (valu - f64)) / (N_SIMD * CLK * us * 1e-6) if valu and us and f64 is not None else None
        hf = hbm / (us * 1e-6) / HBM if hbm and us else None
        lines.append("| %s | %d | %d | %s | %s | %s | %s | %s | %s | %s | %s | %s | %.1f | %s | %s | %s | %s | %s |" % (
            key[0], key[1], n, "%.0f" % (valu / w) if valu else "-",
            "%.0f / %.0f / %.0f" % (fma / w, add / w, mul / w) if f64 is not None else "-",
            "%.0f" % (c['SQ_INSTS_VALU_INT32'] / w) if 'SQ_INSTS_VALU_INT32' in c else "-",
            "%.0f" % (c['SQ_INSTS_SALU'] / w) if 'SQ_INSTS_SALU' in c else "-", "%.0f" % (c['SQ_INSTS_LDS'] / w) if 'SQ_INSTS_LDS' in c else "-",
            "%.2f" % (c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']) if c.get('SQ_WAVE_CYCLES') else "-",
            "%.0f" % fz if fz is not None else "-", "%.0f" % wz if wz is not None else "-", "%.1f" % (hbm / 1e6) if hbm else "-", alg / 1e6,
            "%.2f" % us if us else "-", "%.3f" % wf if wf else "-", "%.3f" % ef if ef else "-", "%.3f" % vf if vf else "-", "%.4f" % hf if hf else "-"))
        shapes_out.append({"workload": WORKLOAD, "kernel": key[0], "particles_per_launch": key[1], "launches": n,
                           "sq_insts_valu_per_launch": valu, "sq_insts_salu_per_launch": c.get('SQ_INSTS_SALU'),
                           "sq_insts_lds_per_launch": c.get('SQ_INSTS_LDS'), "sq_waves_per_launch": c.get('SQ_WAVES'),
                           "sq_wave_cycles_per_launch": c.get('SQ_WAVE_CYCLES'), "sq_wait_any_per_launch": c.get('SQ_WAIT_ANY'),
                           "sq_insts_valu_fma_f64_per_launch": fma, "sq_insts_valu_add_f64_per_launch": add,
                           "sq_insts_valu_mul_f64_per_launch": mul, "sq_insts_valu_trans_f64_per_launch": c.get('SQ_INSTS_VALU_TRANS_F64'),
                           "sq_insts_valu_int32_per_launch": c.get('SQ_INSTS_VALU_INT32'), "sq_insts_valu_int64_per_launch": c.get('SQ_INSTS_VALU_INT64'),
                           "fetch_size_kb": fz, "write_size_kb": wz, "hbm_bytes_per_launch": hbm,
                           "correction": "2*FETCH_SIZE + WRITE_SIZE (KB -> bytes x1024), MI355X_MICROARCH.md HBM section",
                           "alg_bytes_per_launch": alg, "alg_flops_per_launch": FLOP_PER_UNIT * units, "avg_us_in_pmc_pass": us})
    json.dump({"round": tag, "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 20 --warmup 20 "
               "--streams 1 --no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0 (four passes: SQ counters, fp64 / integer instruction classes, FETCH_SIZE, WRITE_SIZE)",
               "launch_shapes": shapes_out}, open(os.path.join(dst, "%s_merge_pmc.json" % tag), "w"), indent=1)
# ---- twisted proposal: SQ counters per kernel (durations from the un-countered trace_twist pass)
f = one("pmc_twist/*/*_counter_collection.csv")
if f:
    tw = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0]
        if name.startswith('pk_twist') or name.startswith('pk_rank_merge'):
            tw[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for extra in ("pmc_twist_fetch", "pmc_twist_write", "pmc_twist_f64"):       # separate passes: TCC slots; fp64 classes
        fe = one("%s/*/*_counter_collection.csv" % extra)
        if fe:
            for r in csv.DictReader(open(fe)):
                name = r['Kernel_Name'].split('(')[0]
                if (name.startswith('pk_twist') or name.startswith('pk_rank_merge')) and r['Counter_Name'] != 'SQ_WAVES':
                    tw[name][r['Counter_Name']].append(float(r['Counter_Value']))
    tdur = collections.defaultdict(list)
    t = one("trace_twist/*/*_kernel_trace.csv")
    if t:
        for r in csv.DictReader(open(t)):
            tdur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    lines += ["## twisted proposal (M = 1, K = 2048): SQ counters per launch, averaged over the rank events", "",
              "Contracts v3 / v4 in place (leaf x leaf rows priced by code pair, leaf x internal rows by code).  VALU issue fraction as above;",
              "durations from the kernel-trace pass.", "",
              "| kernel | launches | avg us | SQ_WAVES | SQ_INSTS_VALU | VALU / wave | SQ_INSTS_SALU | SQ_INSTS_LDS | wait share of wave cycles | executed fp64 frac of 78.6 TF | HBM MB | HBM frac |",
              "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tw_out = []
    for name in sorted(tw):
        c = {k: sum(v) / len(v) for k, v in tw[name].items()}
        us = sum(tdur[name]) / len(tdur[name]) if tdur.get(name) else None
        fz, wz = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
        hbm = (2.0 * fz + wz) * 1024.0 if fz is not None and wz is not None else None
        fma, add, mul = c.get('SQ_INSTS_VALU_FMA_F64'), c.get('SQ_INSTS_VALU_ADD_F64'), c.get('SQ_INSTS_VALU_MUL_F64')
        ef = (2.0 * fma + add + mul) * 64.0 / (us * 1e-6) / FP64_PEAK if None not in (fma, add, mul) and us else None
        lines.append("| %s | %d | %s | %.3g | %.3g | %.0f | %.3g | %.3g | %.2f | %s | %s | %s |" % (
            name, len(tw[name]['SQ_WAVES']), "%.2f" % us if us else "-", c['SQ_WAVES'], c['SQ_INSTS_VALU'], c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1),
            c['SQ_INSTS_SALU'], c['SQ_INSTS_LDS'], c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1),
            "%.3f" % ef if ef else "-", "%.1f" % (hbm / 1e6) if hbm is not None else "-",
            "%.3f" % (hbm / (us * 1e-6) / HBM) if hbm is not None and us else "-"))
        tw_out.append({"workload": WORKLOAD, "kernel": name, "M": 1, "particles": 2048, "launches": len(tw[name]['SQ_WAVES']),
                       "sq_insts_valu_per_launch": c.get('SQ_INSTS_VALU'), "sq_waves_per_launch": c.get('SQ_WAVES'),
                       "sq_insts_salu_per_launch": c.get('SQ_INSTS_SALU'), "sq_insts_lds_per_launch": c.get('SQ_INSTS_LDS'),
                       "sq_insts_valu_fma_f64_per_launch": fma, "sq_insts_valu_add_f64_per_launch": add, "sq_insts_valu_mul_f64_per_launch": mul,
                       "fetch_size_kb": fz, "write_size_kb": wz, "hbm_bytes_per_launch": hbm,
                       "avg_us_in_trace_pass": us, "note": "averages over the N-1 rank events of a sweep (their launch sizes differ)"})
    lines.append("")
    json.dump({"round": tag, "command": "rocprofv3 --kernel-trace --pmc <SQ counters> -- python3 bench.py --twisting --M 1 --steps 3 --warmup 1 --streams 1 "
               "--no-cpu-baseline --no-parity --no-vi-step --min-timed-ms 0", "kernels": tw_out}, open(os.path.join(dst, "%s_twist_pmc.json" % tag), "w"), indent=1)
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
