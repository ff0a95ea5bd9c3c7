"""Determinism soak on one MI355X: plain sweeps and batched sweeps (6 per launch set) interleaved on three contexts
each, lazy nodes (the default); every seed must give the same bits every time and in either form."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phylo_amd import _ffi, model as M  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402

g = load_dataset('primate_data')['genome']
N, S, _ = g.shape
Q = M.get_Q(M.init_y_q())
pi = M.get_stationary_probs(np.zeros(4) + .25)
lam = np.full(N - 1, 10.0)
K, G = 2048, 6


def make(k):
    c = _ffi.Context(k, N, S)
    c.set_leaves(g)
    c.set_model(Q, pi, lam, lam)
    return c


plain = [make(K) for _ in range(3)]
batched = [make(K * G) for _ in range(3)]
ref, bad, n = {}, 0, 0
t0 = time.time()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for rep in range(reps):
    for i, c in enumerate(plain):
        c.sweep_async((i + rep) % 7)
    for i, c in enumerate(batched):
        c.sweep_batch_async([(i + rep + j) % 7 for j in range(G)])
    for i, c in enumerate(plain):
        out = c.sweep_fetch()
        key = (out['logZ'], out['log_weights'].tobytes(), out['ancestors'].tobytes())
        bad += ref.setdefault((i + rep) % 7, key) != key
        n += 1
    for i, c in enumerate(batched):
        out = c.sweep_fetch()
        lz = c.sweep_fetch_logz(G)
        for j in range(G):
            sl = slice(j * K, (j + 1) * K)
            key = (lz[j], np.ascontiguousarray(out['log_weights'][:, sl]).tobytes(),
                   np.ascontiguousarray(out['ancestors'][:, sl]).tobytes())
            bad += ref.setdefault((i + rep + j) % 7, key) != key
            n += 1
print('soak: %d sweeps in %.1f s, mismatches %d, logZ by seed %s' % (n, time.time() - t0, bad, {k: v[0] for k, v in sorted(ref.items())}))
sys.exit(1 if bad else 0)
