import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from phylo_amd import _ffi, model as M
from phylo_amd.datasets import load_dataset
g = load_dataset('primate_data')['genome']; N,S,_ = g.shape
Q = M.get_Q(M.init_y_q()); pi = M.get_stationary_probs(np.zeros(4)+.25); lam = np.full(N-1, 10.0)
K = 2048
for ns in (1,2,3,4):
    ctxs = []
    for i in range(ns):
        c = _ffi.Context(K, N, S); c.set_leaves(g); c.set_model(Q, pi, lam, lam); ctxs.append(c)
    for w in range(6): ctxs[w % ns].sweep_async(1000+w)
    for c in ctxs: c.synchronize()
    steps = 60
    t0 = time.perf_counter()
    for s in range(steps): ctxs[s % ns].sweep_async(s)
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print('streams', ns, 'ms/step %.4f' % (dt/steps*1e3), 'units/s %.3e' % (K*S*(N-1)*steps/dt), 'frac of 8TB/s %.3f' % (96.0*K*S*(N-1)*steps/dt/8e12), flush=True)
    z = [c.sweep_fetch(arrays=False)['logZ'] for c in ctxs]
    print('  logZ', z)
    for c in ctxs: c.close()
