#!/usr/bin/env python3
"""bench.py -- particle-site-likelihoods/sec + |delta log Z-hat| of the CSMC sweep on MI355X (BASELINE.json metric).

A "step" is one full sweep (N-1 rank events: draws, transition matrices, resampling, Felsenstein
merges, weights, log Z-hat) over the alignment already resident in HBM.  Default workload: primate.p
(N=12, S=898), jcmodel=false initial Q (GTR-init), K=2048 particles per GPU.

What the line reports:
  value / ms_per_step   whole-job throughput of the timed region: independent sweeps (own seed, own resampling, own log
                        Z-hat, each bit-identical to the sweep run alone) issued up to ten per set of launches
                        (phylo_sweep_batch_async) on three contexts in flight.  The K steps are timed `repeats` times
                        (>= 100 ms in all) and the MEDIAN repetition is reported.
  t_sweep_ms            device time of ONE sweep alone (hipEvents on its stream), median of >= 20 after 3 warm-ups: the
                        reference's own usage (one evaluation sweep per epoch; training steps are sequential).
  delta_logZ_max, ancestors_equal   seeds 0..9 at the bench's K against the C oracle (the |delta log Z-hat| half of the metric).
  roofline              the dominant kernel (the Felsenstein merge) priced by WORK: frac = achieved / peak with achieved =
                        60 flop x particle-site-likelihoods of one launch / launch duration (SURVEY 8d's algorithmic flops:
                        2 x (16 mul + 12 add) + 4 mul per unit; duration measured live with kernel-stamped HIP events) and peak =
                        78.6 TFLOP/s, the vector fp64 rate (1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz = half the guide's 157.3 TFLOP/s
                        fp32 vector peak; MFMA does not apply to 4x4 contractions).  Beside it, from the committed rocprofv3 --pmc
                        run at this launch shape (profiles/rNN_merge_pmc.json): executed_fp64 (what the SQ_INSTS_VALU_*_F64 counters
                        say was executed: leaf rows are table look-ups, so fewer flops are executed than the algorithm counts),
                        valu_issue_occupancy (a DIAGNOSTIC, not a roofline: it rises when instructions are added), traffic / hbm_frac
                        (HBM bytes from FETCH_SIZE / WRITE_SIZE: with lazy nodes ~2 % of the 96 B/unit).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  No PyTorch is imported: the ranks are joined through RCCL inside
libphylo_hip (phylo_comm_init); the 128-byte RCCL id travels through a file rendezvous on this node.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from phylo_amd import _ffi  # noqa: E402
from phylo_amd import model as M  # noqa: E402
from phylo_amd.datasets import load_dataset, synthetic_alignment  # noqa: E402
from phylo_amd.rendezvous import FileSync, exchange_comm_id  # noqa: E402

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_SIMD = 256 * 4           # 256 CUs x 4 SIMD-32 (MI355X_MICROARCH.md)
CLK_HZ = 2.4e9             # max clock; in-kernel s_memtime / s_memrealtime read 2.36-2.39 GHz on this workload (tests/probe_persist.py)
FP64_PEAK_TFLOPS = N_SIMD * 16 * 2 * CLK_HZ / 1e12   # 78.6: vector fp64 (half the guide's 157.3 TFLOP/s fp32 vector peak)
FLOP_PER_UNIT = 60.0       # 2 x (16 mul + 12 add) + 4 mul per particle-site-likelihood (the log and the dot product not counted)
FLOP_PER_LOOKAHEAD_UNIT = 80.0   # SURVEY 8d, twisted proposal: the merge + the pi dot product + the log-reduce of a look-ahead unit


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=60)
    p.add_argument('--warmup', type=int, default=6)
    p.add_argument('--dataset', default='primate_data')
    p.add_argument('--n_particles', type=int, default=2048, help='particles PER GPU')
    p.add_argument('--jcmodel', default=False, type=lambda x: str(x).lower() == 'true')
    p.add_argument('--synthetic', default=None, help='N,S : synthetic iid-uniform alignment instead of --dataset')
    p.add_argument('--flat', action='store_true',
                   help='replace every row of the alignment by a gap row [1,1,1,1]: near-uniform weights under JC69, children spread over '
                        'all earlier nodes (tests/probe_regimes.py); the opposite extreme of real data')
    p.add_argument('--eager', action='store_true', help='store every node (PHYLO_EAGER_NODES) instead of lazy nodes')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--streams', type=int, default=0,
                   help='independent sweeps kept in flight on separate HIP streams (0 = 3 on one GPU, 1 when sharded)')
    p.add_argument('--batch', type=int, default=0,
                   help='independent sweeps per set of launches (phylo_sweep_batch_async); 0 = the largest divisor of --steps '
                        'up to 20 (plain proposal, small nodes), 1 otherwise')
    p.add_argument('--twisting', action='store_true', help='twisted proposal (vncsmc.py); BASELINE config 2')
    p.add_argument('--M', type=int, default=1, help='sub-samples of the twisted proposal')
    p.add_argument('--params', default=None, help='.npz with Q, pi, lam_l, lam_r (e.g. trained parameters from tests/probe_regimes.py) '
                                                  'instead of the untrained model')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-parity', action='store_true', help='skip the 10-seed comparison with the C oracle')
    p.add_argument('--no-vi-step', action='store_true', help='skip the timing of the VI training steps (SURVEY 8f.1) appended to the line')
    p.add_argument('--min-timed-ms', type=float, default=2000.0, help='repeat the K timed steps until this much time is covered (the K steps alone last a few ms)')
    p.add_argument('--one-launch', action='store_true', help='single sweeps (t_sweep) in the one-launch form (phylo_persist.h)')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU-baseline duration')
    return p.parse_args()


def vi_step_timing(g, K, device=0, steps=12):
    """One VI training step of the widened path: sweep with the graph kept + hand-written reverse pass + Adam, all S
    sites, plain and twisted (M = 1) proposal.  Device times from hipEvents, step time is wall clock; medians of `steps` steps."""
    from phylo_amd import train as T
    N, S, _ = g.shape
    out = {"K": K, "sites": S, "optimizer": "Adam", "unit": "ms", "timer": "hipEvents (sweep, reverse pass); wall clock (step)"}
    for name, nested in (("plain", False), ("twisted_M1", True)):
        v = T.Variables(N, np.log(10.0), False)
        tr = T.Trainer(g, K, v, T.make_optimizer('Adam', 0.01), S, device=device, nested=nested, M=1)
        fw, bw, wall, hostms, who = [], [], [], [], "host"
        try:
            for i in range(steps + 3):
                t0 = time.perf_counter()
                tr.step(np.arange(S), seed=i)
                t1 = time.perf_counter()
                if i >= 3:
                    fw.append(tr.last['raw']['forward_ms']); bw.append(tr.last['raw']['backward_ms']); wall.append((t1 - t0) * 1e3)
                    hostms.append(tr.last['raw'].get('backward_host_ms', 0.0))
                    who = tr.last['raw'].get('backward_lists', 'host')
        finally:
            tr.close()
        # reverse_lists: who builds the reverse pass's integer lists -- "device" (kernels, phylo_revlists_dev.h: lists_host_ms is then the
        # host's wait for ~50 integers) or "host" (twisted proposal, S > 4096: lists_host_ms is the time of the builders)
        out[name] = {"sweep_ms": float(np.median(fw)), "reverse_ms": float(np.median(bw)), "reverse_lists": who,
                     "reverse_lists_host_ms": float(np.median(hostms)), "step_wall_ms": float(np.median(wall))}
    return out


def cpu_baseline(g, Q, pi, lam, jc, K_gpu, seconds, twist_M=0, lam_r=None):
    """The C oracle (reference dataflow, OpenMP) timed on this host's cores on a bounded sample."""
    from oracle import c_oracle as CO
    N, S, _ = g.shape
    # a one-GPU box offers a 16-core CPU share, whatever nproc says; oversubscribing only slows the baseline
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get('PHYLO_CPU_THREADS', '16'))))
    CO.set_threads(cores)
    lam_r = lam if lam_r is None else lam_r

    def one(K, seed):
        if twist_M:
            CO.sweep_twisted(g, Q, pi, lam, lam_r, K, twist_M, seed, jc=jc)
        else:
            CO.sweep(g, Q, pi, lam, lam_r, K, seed, jc=jc)

    t0 = time.perf_counter()
    one(64, 0)
    per_particle = (time.perf_counter() - t0) / 64
    K = int(min(K_gpu, max(64, seconds / max(per_particle, 1e-9))))
    K = 1 << (K.bit_length() - 1)
    n = max(1, int(seconds / (per_particle * K)))
    n = min(n, 200)
    t0 = time.perf_counter()
    for s in range(n):
        one(K, s)
    dt = time.perf_counter() - t0
    units = float(K) * S * ((N - 1) + (twist_M * (N + 1) * N * (N - 1) / 6.0 if twist_M else 0.0)) * n
    return {"value": units / dt, "unit": "particle-site-likelihoods/s", "cores": cores, "kind": "port",
            "sample": "%d sweep(s) of the same alignment at K=%d (oracle/csrc/oracle.c, reference dataflow, %d OpenMP threads, %.1f s)"
                      % (n, K, cores, dt)}


def main():
    a = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs a launcher: python -m torch.distributed.run --nproc-per-node %d bench.py ..." % (a.gpus, a.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, a.gpus))
    if a.synthetic:
        n_taxa, n_sites = (int(v) for v in a.synthetic.split(','))
        d = synthetic_alignment(n_taxa, n_sites)
        wname = "synthetic %dx%d" % (n_taxa, n_sites)
    else:
        d = load_dataset(a.dataset)
        wname = {'primate_data': 'primate.p', 'primate_data_wang': 'primates_small.p'}.get(a.dataset, a.dataset)
    g = d['genome']
    if a.flat:
        g = np.ones_like(g)
        wname += " shape, all-gap rows"
    N, S, _ = g.shape
    Q = M.jc_Q() if a.jcmodel else M.get_Q(M.init_y_q())
    pi = M.get_stationary_probs(np.zeros(4) + 0.25)
    lam = np.full(N - 1, 10.0)                       # branch_prior = log 10 (runner.py:38-41)
    lam_r = lam
    if a.params:
        pz = np.load(a.params)
        Q, pi, lam, lam_r = pz['Q'], pz['pi'], pz['lam_l'], pz['lam_r']
    K_global = a.n_particles * world                  # particles of one sweep (sharded over the ranks)
    # N > 1, last resort and comparison figure: every rank runs its OWN sweeps of --n_particles particles, no exchange at all
    # (PHYLO_BENCH_INDEPENDENT=1 forces it).  The ranks then meet through files (no communicator needed).
    fs = FileSync(rank, world) if world > 1 else None
    independent = False

    ndev = _ffi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    sharded_env = world > 1 or bool(os.environ.get('PHYLO_COMM_FORCE_RCCL'))
    # batching pays where launches are short (small nodes); large nodes fill the GPU with one sweep per launch set
    batch = a.batch if a.batch > 0 else (20 if not a.twisting and S < 8192 else 1)
    if a.batch <= 0:                                  # whole launch sets only: the largest divisor of --steps that is <= the default
        while batch > 1 and a.steps % batch:
            batch -= 1
    if batch > 1 and a.twisting:
        raise SystemExit("--batch needs the plain proposal")
    n_streams = a.streams if a.streams > 0 else (2 if sharded_env and batch > 1 else 3)
    pool_bytes = 32.0 * (N - 1) * a.n_particles * batch * S   # node pool of one context
    while n_streams > 1 and n_streams * pool_bytes > 200e9:
        n_streams -= 1                                # every sweep in flight owns a pool; stay inside 288 GB of HBM
    sharded = world > 1 or bool(os.environ.get('PHYLO_COMM_FORCE_RCCL'))   # the env: rehearse the sharded loop on one rank
    sweep_flags = _ffi.FLAGS_DEFAULT | (_ffi.TWISTING if a.twisting else 0) | (_ffi.EAGER_NODES if a.eager else 0)

    live = []                                             # every context alive, in creation order (closed in reverse: sharers first)

    def make_contexts():
        ctxs = []
        for i in range(n_streams):
            c = _ffi.Context(K_global * batch, N, S, device=local_rank % ndev)
            live.append(c)
            c.set_leaves(g)
            c.set_model(Q, pi, lam, lam_r, jc69_closed_form=a.jcmodel)
            ctxs.append(c)
        if sharded:
            cid = exchange_comm_id(rank, world, _ffi.comm_unique_id if rank == 0 else None)
            ctxs[0].comm_init(rank, world, cid)
            for c in ctxs[1:]:                            # further sweeps in flight: same communicator, one comm stream
                c.comm_share(ctxs[0])
        single = ctxs[0]
        if batch > 1:                                     # one K-particle context: remainder sweeps, single-sweep latency
            single = _ffi.Context(K_global, N, S, device=local_rank % ndev)
            live.append(single)
            single.set_leaves(g)
            single.set_model(Q, pi, lam, lam_r, jc69_closed_form=a.jcmodel)
            if sharded:
                single.comm_share(ctxs[0])
        return ctxs, single

    def close_all():
        while live:
            try:
                live.pop().close()
            except Exception:
                pass

    def barrier():
        if independent:
            fs.barrier()
        else:
            ctx.comm_barrier()

    def gmax(x):
        return fs.max(x) if independent else ctx.comm_max(x)

    ctxs, single, ctx = [], None, None
    first_contact = {"exchange": None, "fallback": None}

    def run(n, seed0):
        nb = n // batch
        if independent:
            seed0 += rank * 1000003                       # every rank its own sweeps
        # device-side exchange: every context has its own slab, flags and stream and no communicator call per rank event, so whole
        # sweeps are issued like on one GPU (PHYLO_BENCH_STEPWISE=1: the rank-event-by-rank-event loop below anyway)
        stepwise = sharded and (first_contact["exchange"] != 'p2p' or bool(os.environ.get('PHYLO_BENCH_STEPWISE')))
        if stepwise:
            # the contexts in flight advance rank event by rank event, so every rank issues the collectives of the shared
            # communicator in the same order; each context carries `batch` independent sweeps (its K = batch * K_global
            # particle indices sharded by contiguous ranges).  One collective per context and rank event lets the
            # contexts drift out of phase, so one computes while the other waits for its all-gather (measured better
            # than one grouped collective for all contexts, PHYLO_BENCH_GROUPED=1, which stalls every context at once)
            for i0 in range(0, nb, n_streams):
                group = ctxs[:min(n_streams, nb - i0)]
                for i, c in enumerate(group):
                    if batch > 1:
                        c.sweep_batch_begin([seed0 + (i0 + i) * batch + j for j in range(batch)], flags=sweep_flags)
                    else:
                        c.sweep_begin(seed0 + i0 + i, flags=sweep_flags, M=a.M)
                for _ in range(N - 1):
                    if os.environ.get('PHYLO_BENCH_GROUPED'):
                        _ffi.sweep_step_group(group)
                    else:
                        for c in group:                   # first halves (lazy nodes: marks, owners' writes, their barrier)
                            c.sweep_step_a()
                        for c in group:                   # second halves: bookkeeping, merge, the all-gather, the scan
                            c.sweep_step()
                for c in group:
                    c.sweep_finish()
        elif batch > 1:
            # `batch` independent sweeps per set of launches, contexts round-robin
            for i in range(nb):
                ctxs[i % n_streams].sweep_batch_async([seed0 + i * batch + j for j in range(batch)], flags=sweep_flags)
        else:
            for s in range(n):
                ctxs[s % n_streams].sweep_async(seed0 + s, flags=sweep_flags, M=a.M)
        if batch > 1:                                     # a remainder runs as single sweeps
            for j in range(n - nb * batch):
                single.sweep_async(seed0 + nb * batch + j, flags=sweep_flags, M=a.M)
            single.synchronize()
        for c in ctxs:
            c.synchronize()

    # untimed: contexts built, every context (and its pool's pages) touched once.  First contact with real links (N > 1): if a form
    # fails on ANY rank (the ranks agree through files, not through the communicator that may be what failed) every rank drops its
    # contexts and tries the next form -- the device-side exchange, then the collective path (PHYLO_P2P=0), then independent sweeps
    # per rank -- and the line says what happened, instead of ending the run without a number.
    forms = ['default'] if world == 1 else (['independent'] if os.environ.get('PHYLO_BENCH_INDEPENDENT') else ['default', 'collective', 'independent'])
    forced_fail = [f for f in os.environ.get('PHYLO_BENCH_FAIL', '').split(',') if f]       # (tests: make a form fail on purpose)
    tried = []
    for form in forms:
        if form == 'collective':
            if first_contact["exchange"] not in ('p2p', None):
                continue                                  # the first form already was the collective path
            os.environ['PHYLO_P2P'] = '0'
        if form == 'independent':
            independent, sharded, K_global = True, False, a.n_particles
            n_streams = a.streams if a.streams > 0 else 3
        failed, why = 0.0, None
        try:
            if form in forced_fail:
                raise _ffi.PhyloError(-1, "forced failure of the '%s' form (PHYLO_BENCH_FAIL)" % form)
            ctxs, single = make_contexts()
            ctx = ctxs[0]
            first_contact["exchange"] = ctx.comm_exchange_kind()
            run(n_streams * batch, a.seed + 2000)
            for c in ctxs:
                c.sweep_fetch(arrays=False)
        except (_ffi.PhyloError, TimeoutError, OSError) as e:
            failed, why = 1.0, "%s: %s" % (type(e).__name__, e)
        if (fs.max(failed) if fs else failed) == 0.0:
            break
        tried.append("%s form failed%s" % (form, (" here: " + why) if why else " on another rank"))
        close_all()
        ctxs, single, ctx = [], None, None
        if world == 1:
            raise SystemExit(why)
    else:
        raise SystemExit("rank %d: no form of the N = %d run worked: %s" % (rank, world, "; ".join(tried)))
    if tried:
        first_contact["fallback"] = "; ".join(tried)
    # Sharded on more than one rank: remote children of a merge are read from a local cache filled once per sweep, or in place over
    # the peer mapping (PHYLO_NO_REMOTE_CACHE=1).  Which is faster depends on the links (on ONE GPU the copy is pure overhead, over
    # xGMI a remote row is otherwise fetched by every launch and XCD): both forms run a few launch sets here, the faster one is kept.
    first_contact["remote_cache"] = None
    if world > 1 and not independent and not a.twisting and 'PHYLO_NO_REMOTE_CACHE' not in os.environ and not os.environ.get('PHYLO_BENCH_NO_CACHE_PROBE'):
        def probe():
            run(n_streams * batch, a.seed + 2100)
            barrier()
            t0 = time.perf_counter()
            for rep in range(3):
                run(n_streams * batch, a.seed + 2200 + rep)
            barrier()
            return gmax(time.perf_counter() - t0) / 3
        t_on = probe()
        used, cap = ctx.debug_remote_cache()
        close_all()
        os.environ['PHYLO_NO_REMOTE_CACHE'] = '1'
        ctxs, single = make_contexts()
        ctx = ctxs[0]
        t_off = probe()
        keep_cache = t_on <= t_off
        if keep_cache:
            close_all()
            del os.environ['PHYLO_NO_REMOTE_CACHE']
            ctxs, single = make_contexts()
            ctx = ctxs[0]
            run(n_streams * batch, a.seed + 2000)
        first_contact["remote_cache"] = {"kept": "cache" if keep_cache else "in place", "ms_per_launch_set_with_cache": t_on * 1e3,
                                         "ms_per_launch_set_in_place": t_off * 1e3, "slots_used_rank0": used, "slots": cap}
    run(-(-max(a.warmup, 0) // batch) * batch, a.seed + 1000)   # W warm-up steps, rounded up to whole launch sets
    # the K steps, timed `repeats` times (each bracketed by the barrier; max over ranks), median reported: one repetition of
    # the default K lasts a few ms, too short to quote alone
    reps_dt = []
    inner = 1
    if independent:                                   # the ranks meet through files (~0.1 ms of skew): time >= 50 ms per bracket
        t0 = time.perf_counter()
        run(a.steps, a.seed)
        inner = max(1, int(gmax(0.05 / max(time.perf_counter() - t0, 1e-6)) + 0.5))
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(inner):
            run(a.steps, a.seed)
        barrier()
        reps_dt.append(gmax(time.perf_counter() - t0) / inner)      # max over ranks (every rank takes the same decision below)
        if sum(reps_dt) * 1e3 >= a.min_timed_ms and len(reps_dt) >= 3:
            break
        if len(reps_dt) >= 4000:
            break
    dt = float(np.median(reps_dt))
    if batch > 1:
        nb = a.steps // batch
        last = single.sweep_fetch(arrays=False) if a.steps % batch else ctxs[(nb - 1) % n_streams].sweep_fetch(arrays=False)
        if a.steps % batch == 0:
            last['logZ'] = float(ctxs[(nb - 1) % n_streams].sweep_fetch_logz(batch)[-1])
    else:
        last = ctxs[(a.steps - 1) % n_streams].sweep_fetch(arrays=False)

    # one sweep at a time on one stream (latency of a single sweep), and the dominant kernel (the
    # Felsenstein merge): average launch duration from HIP events on the ctx stream, nothing else in flight
    single_flags = sweep_flags | (_ffi.ONE_LAUNCH if a.one_launch else 0)
    for s in range(3):                              # 3 warm-ups
        single.sweep_async(a.seed + 3000 + s, flags=single_flags, M=a.M)
    single.synchronize()
    ev_ms = []
    for s in range(24):                             # device time of each sweep alone: hipEvents on the context's stream
        single.sweep_async(a.seed + s, flags=single_flags, M=a.M)
        ev_ms.append(single.sweep_fetch(arrays=False)['stats']['sweep_ms'])
    t_sweep_ms = float(np.median(ev_ms))
    t1 = time.perf_counter()
    for s in range(10):
        single.sweep_async(a.seed + s, flags=single_flags, M=a.M)
    single.synchronize()
    single_ms = (time.perf_counter() - t1) / 10 * 1e3
    # the |delta log Z-hat| half of the metric: seeds 0..9 at this K against the C oracle (identical draws by contract)
    # At N > 1 the same check is the 1 == N contract: the sharded sweep of K_total particles against the oracle's (unsharded) sweep;
    # every rank runs the sweeps (they are collective), rank 0 runs the oracle and compares ITS columns and the global log Z-hat.
    parity = None
    if not a.no_parity and not a.no_cpu_baseline:
        n_seeds = 10 if world <= 2 else 4                 # the oracle's K-replicated core grows with K_total: bounded at N > 2
        worst, same = 0.0, True
        for sd in range(n_seeds):
            single.sweep_async(sd, flags=single_flags, M=a.M)
            out = single.sweep_fetch()
            if rank == 0:
                from oracle import c_oracle as CO
                if a.twisting:
                    ref = CO.sweep_twisted(g, Q, pi, lam, lam_r, K_global, a.M, sd, jc=a.jcmodel)
                else:
                    ref = CO.sweep(g, Q, pi, lam, lam_r, K_global, sd, jc=a.jcmodel)
                worst = max(worst, abs(out['logZ'] - ref['logZ']))
                cols = slice(single.k0, single.k0 + single.K_local)
                same = same and bool(np.array_equal(out['ancestors'], ref['ancestors'][:, cols]))
            barrier()                                     # the other ranks wait for rank 0's oracle HERE, not inside a kernel's flag wait
        parity = {"delta_logZ_max": worst, "ancestors_equal": same, "parity_seeds": n_seeds}
    prof_sweeps = 3
    merge_ms, merge_n = 0.0, 0
    for s in range(prof_sweeps):                    # the launch form of the timed region, one at a time, kernel-stamped events
        if batch > 1:
            ctx.sweep_batch_async([a.seed + s * batch + j for j in range(batch)], flags=sweep_flags | _ffi.TIME_KERNELS)
        else:
            ctx.sweep_async(a.seed + s, flags=sweep_flags | _ffi.TIME_KERNELS, M=a.M)
        st = ctx.sweep_fetch(arrays=False)['stats']
        merge_ms += st['merge_ms']
        merge_n += st['merge_launches']
    bytes_per_launch = 96.0 * ctx.K_local * S       # 2 child reads + 1 parent write, 32 B each, per (particle, site)
    # lazy nodes (plain proposal): the launch stores nothing and runs the row-per-thread form of the merge
    lazy_nodes = not a.twisting and not a.eager and not os.environ.get('PHYLO_EAGER_NODES')
    merge_kernel = "pk_rank_merge_nostore" if lazy_nodes else "pk_rank_merge"
    if a.twisting:
        # the dominant kernel of a twisted sweep is the look-ahead potentials (PHYLO_TIME_KERNELS stamps it): 64 B per look-ahead
        # unit (two child reads, no store), K M S C(N+1,3) units per sweep, averaged over the N-1 launches of a sweep
        merge_kernel = "pk_twist_potentials"
        bytes_per_launch = 64.0 * ctx.K_local * a.M * S * ((N + 1) * N * (N - 1) / 6.0) / (N - 1)
    avg_s = merge_ms / merge_n * 1e-3
    alg_equiv = bytes_per_launch / avg_s / 1e9
    # counters of the committed rocprofv3 --pmc runs at THIS launch shape (they cannot be read inside the run)
    pmc = None
    if a.twisting:
        for tpath in sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', 'r*_twist_pmc.json')), reverse=True):
            try:
                for e in json.load(open(tpath)).get('kernels', []):
                    if e.get('workload') == wname and e.get('kernel') == merge_kernel and e.get('M') == a.M and e.get('particles') == ctx.K_local:
                        pmc = dict(e, source=os.path.basename(tpath))
                        break
            except Exception:
                pass
            if pmc:
                break
    for tpath in ([] if a.twisting else sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', 'r*_merge_pmc.json')), reverse=True)):
        try:
            for e in json.load(open(tpath)).get('launch_shapes', []):
                if e.get('workload') == wname and e.get('particles_per_launch') == ctx.K_local and e.get('kernel') == merge_kernel:
                    pmc = dict(e, source=os.path.basename(tpath))
                    break
        except Exception:
            pass
        if pmc:
            break
    units_per_launch = float(ctx.K_local) * S
    flops_per_launch = FLOP_PER_UNIT * units_per_launch
    if a.twisting:
        units_per_launch = float(ctx.K_local) * a.M * S * ((N + 1) * N * (N - 1) / 6.0) / (N - 1)   # averaged over the rank events
        flops_per_launch = FLOP_PER_LOOKAHEAD_UNIT * units_per_launch
    achieved = flops_per_launch / avg_s / 1e12
    roof = {"kernel": merge_kernel, "avg_launch_us": avg_s * 1e6, "particles_per_launch": ctx.K_local, "units_per_launch": units_per_launch,
            "bound": "valu", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
            "flop_frac": achieved / FP64_PEAK_TFLOPS,
            "basis": ("%.0f algorithmic flop per %s (SURVEY 8d) x units per launch / live launch duration, against the vector fp64 peak "
                      "(1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz); MFMA does not apply (4x4 contractions)"
                      % ((FLOP_PER_LOOKAHEAD_UNIT, "look-ahead unit") if a.twisting else (FLOP_PER_UNIT, "particle-site-likelihood"))),
            "alg_bytes_per_launch": bytes_per_launch, "alg_equiv_GBps": alg_equiv,
            "alg_equiv_note": "bytes of the reference's dataflow per unit x units / duration: an algorithmic-equivalent rate, NOT HBM traffic "
                              "(lazy nodes store nothing; children are 1-byte codes or cache-resident)",
            "traffic": None, "hbm_frac": None, "executed_fp64_TFLOPs": None, "executed_fp64_frac": None, "valu_issue_occupancy": None,
            "pmc_source": pmc['source'] if pmc else None}
    if pmc:
        if pmc.get('hbm_bytes_per_launch') is not None:
            roof["traffic"] = pmc['hbm_bytes_per_launch']
            roof["hbm_GBps"] = pmc['hbm_bytes_per_launch'] / avg_s / 1e9
            roof["hbm_frac"] = roof["hbm_GBps"] / HBM_PEAK_GBPS
        fma, add, mul = (pmc.get('sq_insts_valu_%s_f64_per_launch' % k) for k in ('fma', 'add', 'mul'))
        if None not in (fma, add, mul):
            roof["executed_fp64_TFLOPs"] = (2.0 * fma + add + mul) * 64.0 / avg_s / 1e12
            roof["executed_fp64_frac"] = roof["executed_fp64_TFLOPs"] / FP64_PEAK_TFLOPS
            if pmc.get('sq_insts_valu_per_launch') is not None:
                f64 = fma + add + mul
                roof["valu_issue_occupancy"] = (4.0 * f64 + 2.0 * (pmc['sq_insts_valu_per_launch'] - f64)) / (N_SIMD * CLK_HZ * avg_s)
                roof["valu_issue_note"] = ("diagnostic only: fp64 instructions priced at 4 cycles, every other VALU instruction at 2 (a lower "
                                           "bound: shifts, compares and DPP moves measure 4); it goes UP when instructions are added")
        if roof["frac"] > 1.0 and roof["executed_fp64_frac"] is not None:
            # the algorithmic count prices arithmetic the kernel does not execute (coded leaves go through tables: contracts v3 / v4 of
            # the twisted proposal): above 1 it is no fraction of any peak -- the executed fp64 rate is reported instead
            roof.update({"alg_flop_frac": roof["frac"], "achieved": roof["executed_fp64_TFLOPs"], "frac": roof["executed_fp64_frac"],
                         "frac_note": "executed fp64 (SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 of the committed counter pass) / live duration: the "
                                      "algorithmic flops / duration (alg_flop_frac) exceed the peak because table look-ups replace most of them"})
        if roof["hbm_frac"] is not None and roof["hbm_frac"] > 0.5:        # eager nodes on large rows: the store stream binds
            roof.update({"bound": "hbm", "achieved": roof["hbm_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": roof["hbm_frac"]})

    # N > 1, sharded: the same steps once more as independent sweeps per rank (own contexts, no communicator): what N GPUs do
    # without any exchange, next to which the sharded value shows the price of the global resampling
    independent_rate = None
    if world > 1 and not independent and not a.twisting and not os.environ.get('PHYLO_BENCH_NO_INDEPENDENT'):
        Ks, sh, ns, cx, sg, c0, n_live = K_global, sharded, n_streams, ctxs, single, ctx, len(live)
        try:
            independent, sharded, K_global, n_streams = True, False, a.n_particles, (a.streams if a.streams > 0 else 3)
            ctxs, single = make_contexts()
            ctx = ctxs[0]
            run(n_streams * batch, a.seed + 2000)
            t0 = time.perf_counter()
            run(a.steps, a.seed)
            inner = max(1, int(gmax(0.05 / max(time.perf_counter() - t0, 1e-6)) + 0.5))
            barrier()
            t0 = time.perf_counter()
            for _ in range(inner):
                run(a.steps, a.seed)
            barrier()
            dti = gmax(time.perf_counter() - t0) / inner
            independent_rate = float(a.n_particles) * S * (N - 1) * a.steps * world / dti
        except (_ffi.PhyloError, TimeoutError, OSError):
            independent_rate = None
        while len(live) > n_live:
            live.pop().close()
        independent, sharded, K_global, n_streams, ctxs, single, ctx = False, sh, Ks, ns, cx, sg, c0
    if rank == 0:
        units_per_step = float(K_global) * S * (N - 1)
        if a.twisting:                             # + K M S C(N+1,3) look-ahead merges
            units_per_step += float(K_global) * a.M * S * ((N + 1) * N * (N - 1) / 6.0)
        line = {
            "metric": "particle-site-likelihoods/sec", "value": units_per_step * a.steps / dt * (world if independent else 1),
            "unit": "particle-site-likelihoods/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if a.synthetic else "primate.p alignment (real sites), %s model parameters" % ("trained (%s)" % os.path.basename(a.params) if a.params else "untrained"),
            "config": {"workload": "%s N=%d S=%d, %s, K=%d per GPU (K_total=%d), lambda=10, full sweep of %d rank events"
                                   % (wname, N, S, ("JC69" if a.jcmodel else "GTR-init (jcmodel=false)") + (" + twisting M=%d" % a.M if a.twisting else ""),
                                      a.n_particles, K_global, N - 1),
                       "parallelism": ("independent sweeps of K=%d particles on each of %d GPUs, no exchange" % (a.n_particles, world)) if independent
                                      else "particles sharded over %d GPU(s), global resampling" % world,
                       "sweeps_per_launch_set": batch, "contexts_in_flight": n_streams,
                       "sweeps_in_flight": n_streams * batch, "particles_in_flight": n_streams * batch * K_global},
            "timed_region": {"repeats": len(reps_dt), "ms_total": sum(reps_dt) * 1e3, "ms_per_step_min": min(reps_dt) / a.steps * 1e3,
                             "ms_per_step_max": max(reps_dt) / a.steps * 1e3, "reported": "median repetition"},
            "t_sweep_ms": t_sweep_ms,
            "t_sweep": {"n": len(ev_ms), "min_ms": min(ev_ms), "max_ms": max(ev_ms), "timer": "hipEvents on the sweep's stream",
                        "form": "one launch (phylo_persist.h)" if a.one_launch else "launches per rank event",
                        "units_per_s": units_per_step / world / (t_sweep_ms * 1e-3),
                        "flop_frac": (FLOP_PER_UNIT * float(K_global) * S * (N - 1) + (FLOP_PER_LOOKAHEAD_UNIT * (units_per_step - float(K_global) * S * (N - 1))))
                                     / world / (t_sweep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
            "single_sweep_wall_ms": single_ms,
            "log_Z": last['logZ'],
            "roofline": roof,
        }
        if parity is not None:
            line.update(parity)
        if world > 1 or sharded:
            indep = None
            if independent_rate is not None:
                indep = {"value": independent_rate, "unit": "particle-site-likelihoods/s",
                         "what": "the same %d steps as %d independent sweeps of K=%d per rank, no exchange: the N x one-GPU figure the sharded value is to "
                                 "be read against" % (a.steps, world, a.n_particles)}
            # first contact with more than one device: what the run actually did, next to the one-GPU figures it should match
            n_coll = 0 if first_contact["exchange"] == 'p2p' else (N - 1) * (2 if not a.eager and not a.twisting else 1) - (1 if not a.eager and not a.twisting else 0)
            ref_us = None
            for tpath in sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', 'r*_merge_pmc.json')), reverse=True):
                try:
                    for e in json.load(open(tpath)).get('launch_shapes', []):
                        if e.get('workload') == wname and e.get('particles_per_launch') == ctx.K_local and e.get('kernel') == merge_kernel:
                            ref_us = e.get('avg_us_in_pmc_pass')
                except Exception:
                    pass
                if ref_us:
                    break
            # where the particles of a sweep live: a context's K = batch x K_total particle indices are sharded by contiguous ranges, so with
            # batch >= N a sweep's K_total particles lie on one rank or straddle a boundary; only straddling sweeps merge remote children
            kl_ctx = K_global * batch // max(world, 1)
            straddling = sum(1 for gi in range(batch) if (gi * K_global) // kl_ctx != ((gi + 1) * K_global - 1) // kl_ctx)
            line["multi_gpu"] = {"form": "independent sweeps per rank" if independent else "one sweep's particles sharded over the ranks",
                                 "sweeps_per_launch_set": batch, "sweeps_whose_particles_straddle_ranks": None if independent else straddling,
                                 "layout_note": None if independent else
                                 "every rank scans and exchanges the weights of all %d sweeps of a launch set; a sweep whose K_total particles lie on one "
                                 "rank reads no remote child (with --batch 1 every sweep is spread over all ranks)" % batch,
                                 "independent_sweeps": indep,
                                 "exchange": first_contact["exchange"], "fallback": first_contact["fallback"],
                                 "remote_children": first_contact.get("remote_cache"),
                                 "collective_calls_per_sweep_and_rank": n_coll,
                                 "merge_kernel": merge_kernel, "merge_avg_launch_us_rank0": avg_s * 1e6,
                                 "merge_avg_launch_us_one_gpu_profile": ref_us,
                                 "note": "merge duration above the one-GPU figure at the same particles per launch = remote child rows are not "
                                         "served from this GPU's caches (DESIGN.md section 5); delta_logZ_max / ancestors_equal are the "
                                         "1 == N contract against the unsharded oracle sweep of K_total particles"}
        if world == 1 and not a.no_vi_step and not a.synthetic:
            line["vi_step"] = vi_step_timing(g, a.n_particles, device=local_rank % ndev)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(g, Q, pi, lam, a.jcmodel, a.n_particles, a.cpu_seconds, a.M if a.twisting else 0, lam_r)
        print(json.dumps(line), flush=True)
    close_all()                                       # (sharers before the owner of the communicator)


if __name__ == '__main__':
    main()
