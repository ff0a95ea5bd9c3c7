"""Random (N, S, K, model, flags, form) draws, GPU against the C oracle, bit for bit; wider ranges than the test suite's 60 draws
(up to 70 taxa: the 8 / 16 / 32 / 64-lane and the wave-per-particle bookkeeping; batched groups, small and large; twisting; one-launch form; flat
weights; site tiles from 64 sites to the default, rows of one to nine tiles).  python tests/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO, cpu_ref as O     # noqa: E402  (a checker, like tests/)
from phylo_amd import _ffi                          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())


t0, n, kinds, last = time.time(), 0, {}, time.time()
while time.time() - t0 < budget:
    N = int(rng.choice([2, 3, 5, 9, 12, 16, 17, 27, 32, 33, 40, 50, 64, 65, 70]))
    S = int(rng.choice([1, 7, 63, 64, 65, 255, 256, 257, 600, 1025, 2049, 4100]))
    T = int(rng.choice([0, 0, 64, 128, 448, 1024]))           # contract v5's site tile (0: the default); the oracle takes the same
    mode = str(rng.choice(['plain', 'plain', 'batched', 'batched_large', 'twist', 'one_launch', 'eager', 'flat']))
    Kg = int(rng.choice([1, 3, 16, 50, 129, 256, 700]))
    G = int(rng.choice([2, 3, 5])) if mode == 'batched' else 1
    if mode == 'batched_large':                               # launch sets of >= 8192 particles: 8-lane bookkeeping, sorted prologue
        N, S = int(rng.choice([5, 9, 12, 16])), int(rng.choice([7, 64, 257, 600]))
        G, Kg = int(rng.choice([9, 13, 17])), int(rng.choice([700, 1000, 1024]))
    if mode == 'twist':
        N, Kg = min(N, 12), min(Kg, 50)
    if S > 2048:
        N, Kg = min(N, 17), min(Kg, 50)                       # (the oracle's K-replicated core)
    M = int(rng.choice([1, 2, 5])) if mode == 'twist' else 1
    jc = bool(rng.integers(0, 2))
    q1 = bool(rng.integers(0, 2))
    if mode == 'flat':
        g = np.ones((N, S, 4))
    else:
        codes = rng.integers(0, 5, size=(N, S))
        g = np.zeros((N, S, 4))
        for a in range(4):
            g[..., a] = (codes == a) | (codes == 4)
        if rng.integers(0, 6) == 0:
            g[rng.integers(0, N), rng.integers(0, S)] = rng.uniform(0.1, 1.0, size=4)      # a generic row: no leaf codes
    if jc:
        Q = O.jc_Q()
    else:
        y = rng.normal(size=(4, 4)) * 0.4
        np.fill_diagonal(y, 0.0)
        Q = O.get_Q(y)
    p = np.exp(rng.normal(size=4) * 0.3)
    pi = (p / p.sum())[None, :]
    lam_l = np.exp(rng.normal(size=N - 1) * 0.3 + 2.0)
    lam_r = np.exp(rng.normal(size=N - 1) * 0.3 + 2.0)
    flags = (_ffi.QUIRK_Q1_RAW_Q if q1 else 0) | (_ffi.EAGER_NODES if mode == 'eager' else 0) | (_ffi.ONE_LAUNCH if mode == 'one_launch' else 0)
    oflags = O.QUIRK_Q1_RAW_Q if q1 else 0
    seeds = [int(rng.integers(0, 2 ** 40)) for _ in range(G)]
    what = "N=%d S=%d T=%d mode=%s G=%d Kg=%d M=%d jc=%s q1=%s" % (N, S, T, mode, G, Kg, M, jc, q1)
    CO.set_site_tile(T)
    with _ffi.Context(G * Kg, N, S) as ctx:
        ctx.set_site_tile(T)
        ctx.set_leaves(g)
        ctx.set_model(Q, pi, lam_l, lam_r, jc69_closed_form=jc)
        if mode == 'twist':
            out = ctx.sweep(seeds[0], flags=flags | _ffi.TWISTING, M=M)
            refs = [CO.sweep_twisted(g, Q, pi, lam_l, lam_r, Kg, M, seeds[0], jc=jc)]
            logz = [out['logZ']]
        elif G == 1:
            out = ctx.sweep(seeds[0], flags=flags)
            refs = [CO.sweep(g, Q, pi, lam_l, lam_r, Kg, seeds[0], flags=oflags, jc=jc)]
            logz = [out['logZ']]
        else:
            ctx.sweep_batch_async(seeds, flags=flags)
            out = ctx.sweep_fetch()
            logz = list(ctx.sweep_fetch_logz(G))
            refs = [CO.sweep(g, Q, pi, lam_l, lam_r, Kg, s, flags=oflags, jc=jc) for s in seeds]
    for key in ('log_weights', 'log_likelihood'):
        assert bits(out[key], np.concatenate([r[key] for r in refs], axis=1)), (what, key)
    assert np.array_equal(out['ancestors'], np.concatenate([r['ancestors'] for r in refs], axis=1)), what
    assert np.array_equal(out['merges'], np.concatenate([r['merges'] for r in refs], axis=1)), what
    assert all(bits(a, b['logZ']) for a, b in zip(logz, refs)), what
    n += 1
    kinds[mode] = kinds.get(mode, 0) + 1
    if time.time() - last > 30:
        print("  ... %d configurations so far (%s)" % (n, what), flush=True)
        last = time.time()
print("%d random configurations bit-exact against the C oracle in %.0f s: %s" % (n, time.time() - t0, kinds))
