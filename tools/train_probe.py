"""Timing of one VI training step (sweep with the graph kept + reverse pass + host update) and a short ELBO
trajectory.  python tools/train_probe.py [--K 2048] [--sites 898] [--steps 20] [--epochs 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylo_amd import train as T                    # noqa: E402
from phylo_amd.datasets import load_dataset         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dataset', default='primate_data')
ap.add_argument('--K', type=int, default=2048)
ap.add_argument('--sites', type=int, default=0)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--jcmodel', action='store_true')
ap.add_argument('--nested', action='store_true')
ap.add_argument('--phases', action='store_true')
ap.add_argument('--M', type=int, default=1)
a = ap.parse_args()

genome = load_dataset(a.dataset)['genome']
N, S, _ = genome.shape
B = a.sites or S
v = T.Variables(N, np.log(10.0), a.jcmodel)
tr = T.Trainer(genome, a.K, v, T.make_optimizer('Adam', 0.01), B, nested=a.nested, M=a.M)
rng = np.random.default_rng(0)
fw, bw, wall, hostms = [], [], [], []
phases = {}
if a.phases:                      # wall time of every host call of a step (Trainer.gradients, taken apart)
    orig = {}
    for name in ('set_model', 'sweep_async', 'sweep_fetch', 'sweep_backward', 'set_leaves'):
        fn = getattr(tr.ctx, name)
        def timed(*args, _fn=fn, _name=name, **kw):
            t = time.perf_counter()
            out = _fn(*args, **kw)
            phases.setdefault(_name, []).append((time.perf_counter() - t) * 1e3)
            return out
        setattr(tr.ctx, name, timed)
for i in range(a.steps + 3):
    sites = np.sort(rng.permutation(S)[:B])
    if i == 3:
        phases.clear()
    t0 = time.perf_counter()
    tr.step(sites, seed=i)
    t1 = time.perf_counter()
    if i >= 3:
        fw.append(tr.last['raw']['forward_ms'])
        bw.append(tr.last['raw']['backward_ms'])
        hostms.append(tr.last['raw'].get('backward_host_ms', 0.0))
        wall.append((t1 - t0) * 1e3)
if a.phases:
    print(json.dumps({'host_call_ms': {k: float(np.mean(v)) for k, v in phases.items()}}))
print(json.dumps({'dataset': a.dataset, 'nested': a.nested, 'M': a.M, 'K': a.K, 'N': N, 'sites': B, 'steps': a.steps,
                  'forward_ms': float(np.mean(fw)), 'backward_ms': float(np.mean(bw)), 'backward_host_ms': float(np.mean(hostms)), 'step_wall_ms': float(np.mean(wall)),
                  'step_wall_ms_min': float(np.min(wall)), 'last_logZ': tr.last['logZ']}))
tr.close()
