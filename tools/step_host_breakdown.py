"""Where the host time of a VI training step goes (wall clock per phase, median of `steps` steps): python tools/step_host_breakdown.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylo_amd import _ffi, train as T  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402

g = load_dataset('primate_data')['genome']
N, S, _ = g.shape
K = 2048
v = T.Variables(N, np.log(10.0), False)
tr = T.Trainer(g, K, v, T.make_optimizer('Adam', 0.01), S)
sites = np.arange(S)
names = ['evaluate', 'set_leaves?', 'set_model', 'sweep_async', 'backward', 'fetch', 'chain_rules', 'adam']
acc = {n: [] for n in names}
tot = []
for i in range(40):
    t = [time.perf_counter()]
    Q, pi, ll, lr = v.evaluate(); t.append(time.perf_counter())
    if tr._sites is None or not np.array_equal(sites, tr._sites):
        tr.ctx.set_leaves(tr.genome[:, sites, :]); tr._sites = sites.copy()
    t.append(time.perf_counter())
    tr.ctx.set_model(Q, pi, ll, lr, jc69_closed_form=v.jc); t.append(time.perf_counter())
    tr.ctx.sweep_async(i, tr.flags, tr.M); t.append(time.perf_counter())
    raw = tr.ctx.sweep_backward(); t.append(time.perf_counter())
    out = tr.ctx.sweep_fetch(arrays=False); t.append(time.perf_counter())
    grads = T.chain_rules(v, Q, pi, ll, lr, raw); t.append(time.perf_counter())
    tr.opt.apply(v, grads); t.append(time.perf_counter())
    if i >= 8:
        for n, a, b in zip(names, t[:-1], t[1:]):
            acc[n].append((b - a) * 1e6)
        tot.append((t[-1] - t[0]) * 1e6)
for n in names:
    print("%-14s %7.1f us" % (n, float(np.median(acc[n]))))
print("%-14s %7.1f us   (device: sweep %.1f us, reverse %.1f us)" % ('step', float(np.median(tot)), out['stats']['sweep_ms'] * 1e3, raw['backward_ms'] * 1e3))
tr.close()
