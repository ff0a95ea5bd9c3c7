import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from phylo_amd import _ffi, model as M
from phylo_amd.datasets import load_dataset
g = load_dataset('primate_data')['genome']; N,S,_ = g.shape
Q = M.get_Q(M.init_y_q()); pi = M.get_stationary_probs(np.zeros(4)+.25); lam = np.full(N-1, 10.0)
K = 2048
ctxs = []
for i in range(3):
    c = _ffi.Context(K, N, S); c.set_leaves(g); c.set_model(Q, pi, lam, lam); ctxs.append(c)
ref = {}
bad = 0
t0 = time.time()
for rep in range(400):
    for i, c in enumerate(ctxs): c.sweep_async((i + rep) % 5)
    for i, c in enumerate(ctxs):
        out = c.sweep_fetch()
        seed = (i + rep) % 5
        key = (out['logZ'], out['log_weights'].tobytes(), out['ancestors'].tobytes())
        if ref.setdefault(seed, key) != key: bad += 1
print('soak: 1200 sweeps in %.1f s, mismatches %d, logZ by seed %s' % (time.time()-t0, bad, {k: v[0] for k, v in ref.items()}))
