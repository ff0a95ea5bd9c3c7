#!/usr/bin/env python3
"""bench.py -- particle-site-likelihoods/sec of the CSMC sweep on MI355X (BASELINE.json metric).

A "step" is one full sweep (N-1 rank events: draws, transition matrices, resampling, Felsenstein
merges, weights, log Z-hat) over the alignment already resident in HBM.  Default workload: primate.p
(N=12, S=898), jcmodel=false initial Q (GTR-init), K=2048 particles per GPU.  Throughput form on one GPU:
independent sweeps (own seed, own resampling, own log Z-hat, each bit-identical to the sweep run alone) are issued
up to ten per set of launches (phylo_sweep_batch_async) on three contexts in flight; `single_sweep_ms` is the
latency of one sweep alone.

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
from phylo_amd.rendezvous import exchange_comm_id  # noqa: E402

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=60)
    p.add_argument('--warmup', type=int, default=6)
    p.add_argument('--dataset', default='primate_data')
    p.add_argument('--n_particles', type=int, default=2048, help='particles PER GPU')
    p.add_argument('--jcmodel', default=False, type=lambda x: str(x).lower() == 'true')
    p.add_argument('--synthetic', default=None, help='N,S : synthetic iid-uniform alignment instead of --dataset')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--streams', type=int, default=0,
                   help='independent sweeps kept in flight on separate HIP streams (0 = 3 on one GPU, 1 when sharded)')
    p.add_argument('--batch', type=int, default=0,
                   help='independent sweeps per set of launches (phylo_sweep_batch_async); 0 = the largest divisor of --steps '
                        'up to 10 (plain proposal, small nodes), 1 otherwise')
    p.add_argument('--twisting', action='store_true', help='twisted proposal (vncsmc.py); BASELINE config 2')
    p.add_argument('--M', type=int, default=1, help='sub-samples of the twisted proposal')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU-baseline duration')
    return p.parse_args()


def cpu_baseline(g, Q, pi, lam, jc, K_gpu, seconds, twist_M=0):
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
    def one(K, seed):
        if twist_M:
            CO.sweep_twisted(g, Q, pi, lam, lam, K, twist_M, seed, jc=jc)
        else:
            CO.sweep(g, Q, pi, lam, lam, K, seed, jc=jc)

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
    N, S, _ = g.shape
    Q = M.jc_Q() if a.jcmodel else M.get_Q(M.init_y_q())
    pi = M.get_stationary_probs(np.zeros(4) + 0.25)
    lam = np.full(N - 1, 10.0)                       # branch_prior = log 10 (runner.py:38-41)
    K_global = a.n_particles * world

    ndev = _ffi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    sharded_env = world > 1 or bool(os.environ.get('PHYLO_COMM_FORCE_RCCL'))
    # batching pays where launches are short (small nodes); large nodes fill the GPU with one sweep per launch set
    batch = a.batch if a.batch > 0 else (10 if not a.twisting and S < 8192 else 1)
    if a.batch <= 0:                                  # whole launch sets only: the largest divisor of --steps that is <= the default
        while batch > 1 and a.steps % batch:
            batch -= 1
    if batch > 1 and a.twisting:
        raise SystemExit("--batch needs the plain proposal")
    n_streams = a.streams if a.streams > 0 else (2 if sharded_env and batch > 1 else 3)
    pool_bytes = 32.0 * (N - 1) * a.n_particles * batch * S   # node pool of one context
    while n_streams > 1 and n_streams * pool_bytes > 200e9:
        n_streams -= 1                                # every sweep in flight owns a pool; stay inside 288 GB of HBM
    ctxs = []
    for i in range(n_streams):
        c = _ffi.Context(K_global * batch, N, S, device=local_rank % ndev)
        c.set_leaves(g)
        c.set_model(Q, pi, lam, lam, jc69_closed_form=a.jcmodel)
        ctxs.append(c)
    ctx = ctxs[0]
    sharded = world > 1 or bool(os.environ.get('PHYLO_COMM_FORCE_RCCL'))   # the env: rehearse the sharded loop on one rank
    if sharded:
        cid = exchange_comm_id(rank, world, _ffi.comm_unique_id if rank == 0 else None)
        ctx.comm_init(rank, world, cid)
        for c in ctxs[1:]:                            # further sweeps in flight: same communicator, one comm stream
            c.comm_share(ctx)

    sweep_flags = _ffi.FLAGS_DEFAULT | (_ffi.TWISTING if a.twisting else 0)
    single = ctx
    if batch > 1:                                     # one K-particle context: remainder sweeps, single-sweep latency
        single = _ffi.Context(K_global, N, S, device=local_rank % ndev)
        single.set_leaves(g)
        single.set_model(Q, pi, lam, lam, jc69_closed_form=a.jcmodel)
        if sharded:
            single.comm_share(ctx)

    def run(n, seed0):
        nb = n // batch
        if sharded:
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

    run(n_streams * batch, a.seed + 2000)            # untimed: every context (and its pool's pages) touched once
    run(-(-max(a.warmup, 0) // batch) * batch, a.seed + 1000)   # W warm-up steps, rounded up to whole launch sets
    ctx.comm_barrier()
    t0 = time.perf_counter()
    run(a.steps, a.seed)
    ctx.comm_barrier()
    dt = time.perf_counter() - t0
    dt = ctx.comm_max(dt)                            # max over ranks
    if batch > 1:
        nb = a.steps // batch
        last = single.sweep_fetch(arrays=False) if a.steps % batch else ctxs[(nb - 1) % n_streams].sweep_fetch(arrays=False)
        if a.steps % batch == 0:
            last['logZ'] = float(ctxs[(nb - 1) % n_streams].sweep_fetch_logz(batch)[-1])
    else:
        last = ctxs[(a.steps - 1) % n_streams].sweep_fetch(arrays=False)

    # one sweep at a time on one stream (latency of a single sweep), and the dominant kernel (the
    # Felsenstein merge): average launch duration from HIP events on the ctx stream, nothing else in flight
    for s in range(3):                              # untimed: this context's pages and caches
        single.sweep_async(a.seed + 3000 + s, flags=sweep_flags, M=a.M)
    single.synchronize()
    t1 = time.perf_counter()
    for s in range(10):
        single.sweep_async(a.seed + s, flags=sweep_flags, M=a.M)
    single.synchronize()
    single_ms = (time.perf_counter() - t1) / 10 * 1e3
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
    lazy_nodes = not a.twisting and not os.environ.get('PHYLO_EAGER_NODES')
    merge_kernel = "pk_rank_merge_nostore" if lazy_nodes else "pk_rank_merge"
    avg_s = merge_ms / merge_n * 1e-3
    achieved = bytes_per_launch / avg_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get('workload') == wname and tj.get('K') == ctx.K_local and tj.get('kernel') == merge_kernel:
                traffic = tj.get('hbm_bytes_per_launch')
        except Exception:
            traffic = None

    if rank == 0:
        units_per_step = float(K_global) * S * (N - 1)
        if a.twisting:                             # + K M S C(N+1,3) look-ahead merges
            units_per_step += float(K_global) * a.M * S * ((N + 1) * N * (N - 1) / 6.0)
        line = {
            "metric": "particle-site-likelihoods/sec", "value": units_per_step * a.steps / dt,
            "unit": "particle-site-likelihoods/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if a.synthetic else "primate.p alignment (real sites), untrained model parameters",
            "config": {"workload": "%s N=%d S=%d, %s, K=%d per GPU (K_total=%d), lambda=10, full sweep of %d rank events"
                                   % (wname, N, S, ("JC69" if a.jcmodel else "GTR-init (jcmodel=false)") + (" + twisting M=%d" % a.M if a.twisting else ""),
                                      a.n_particles, K_global, N - 1),
                       "parallelism": "particles sharded over %d GPU(s), global resampling" % world,
                       "sweeps_per_launch_set": batch, "contexts_in_flight": n_streams,
                       "sweeps_in_flight": n_streams * batch},
            "single_sweep_ms": single_ms,
            "log_Z": last['logZ'],
            "roofline": {"bound": "hbm", "kernel": merge_kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "alg_bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_s * 1e6,
                         "sweep_frac_of_peak": (96.0 * units_per_step / world) / (dt / a.steps) / 1e9 / HBM_PEAK_GBPS},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(g, Q, pi, lam, a.jcmodel, a.n_particles, a.cpu_seconds, a.M if a.twisting else 0)
        print(json.dumps(line), flush=True)
    if single is not ctx:
        single.close()
    for c in reversed(ctxs):                          # sharers before the owner of the communicator
        c.close()


if __name__ == '__main__':
    main()
