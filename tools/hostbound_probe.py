import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from phylo_amd import _ffi, model as M
from phylo_amd.datasets import load_dataset
g = load_dataset('primate_data')['genome']; N,S,_ = g.shape
Q = M.get_Q(M.init_y_q()); pi = M.get_stationary_probs(np.zeros(4)+.25); lam = np.full(N-1, 10.0)
K = 2048
for ns in (1,2,3,4,6):
    ctxs = []
    for i in range(ns):
        c = _ffi.Context(K, N, S); c.set_leaves(g); c.set_model(Q, pi, lam, lam); ctxs.append(c)
    for w in range(2*ns): ctxs[w % ns].sweep_async(1000+w)
    for c in ctxs: c.synchronize()
    steps = 60
    t0 = time.perf_counter()
    for s in range(steps): ctxs[s % ns].sweep_async(s)
    t1 = time.perf_counter()
    for c in ctxs: c.synchronize()
    t2 = time.perf_counter()
    print('streams', ns, 'host enqueue per sweep %.1f us' % ((t1-t0)/steps*1e6), 'total per sweep %.1f us' % ((t2-t0)/steps*1e6), flush=True)
    for c in ctxs: c.close()
