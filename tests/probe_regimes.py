#!/usr/bin/env python3
"""Sweeps outside the degenerate regime of the untrained model.

At the reference's initial parameters (lambda = 10, uniform Q) 5-27 of K = 2048 particles survive each resampling, so lazy
nodes write almost nothing and every child row sits in L2.  This tool
  1. trains the model with the product's own VI step (phylo_amd.train: Adam, site minibatches, the reverse pass) for a few
     epochs on primate.p and writes the parameters to an .npz (bench.py --params reads it);
  2. runs the sweep THROUGH THE LIBRARY with untrained and trained parameters, lazy and eager nodes, launches and the one-launch
     form, and reports per configuration: distinct ancestors per rank event (= nodes materialised by lazy nodes), device time
     per sweep (hipEvents), units/s; every configuration is also compared bit for bit with the C oracle on seed 0.
usage: python tests/probe_regimes.py [--epochs 40] [--K 2048] [--out gpurun_out/regimes.json] [--params-out profiles/r02_trained_params.npz]
"""
import argparse
import json
import os
import random
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from phylo_amd import _ffi, model as M  # noqa: E402
from phylo_amd import train as T  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402


def train(g, K, epochs, lr, batch):
    N, S, _ = g.shape
    v = T.Variables(N, np.log(10.0), False)
    tr = T.Trainer(g, K, v, T.make_optimizer('Adam', lr), batch)
    random.seed(0)
    seed = 0
    hist = []
    for ep in range(epochs):
        sites = list(range(S))
        random.shuffle(sites)
        for j in range(S // batch):                      # every slice but the leftover (vcsmc.py:533)
            tr.step(sorted(sites[j * batch:(j + 1) * batch]), seed=seed)
            seed += 1
        hist.append(tr.last['logZ'])
    tr.close()
    return v, hist


def flat_alignment(N, S):
    """All-gap rows ([1,1,1,1], the reference's encoding of '-', runner.py:95-96): under JC69 every site likelihood is 1, so
    the weights differ by prior / proposal terms only and resampling is close to uniform: ~63 % of the particles survive and the
    children of a merge are spread over all earlier nodes -- the opposite extreme of real data."""
    return np.ones((N, S, 4))


def measure(g, K, Q, pi, lam_l, lam_r, flags, reps, check, jc=False):
    N, S, _ = g.shape
    ctx = _ffi.Context(K, N, S)
    ctx.set_leaves(g)
    ctx.set_model(Q, pi, lam_l, lam_r, jc69_closed_form=jc)
    out = ctx.sweep(0, flags=flags)
    if check:
        from oracle import c_oracle as CO
        ref = CO.sweep(g, Q, pi, lam_l, lam_r, K, 0, jc=jc)
        assert np.array_equal(out['ancestors'], ref['ancestors']) and out['logZ'] == ref['logZ'], "differs from the C oracle"
        assert np.array_equal(out['log_weights'].view(np.uint64), ref['log_weights'].view(np.uint64))
    distinct = [int(len(np.unique(a))) for a in out['ancestors']]
    ms = []
    for s in range(reps):
        ctx.sweep_async(s, flags=flags)
        ms.append(ctx.sweep_fetch(arrays=False)['stats']['sweep_ms'])
    ctx.close()
    t = float(np.median(ms))
    return {"K": K, "N": N, "S": S, "t_sweep_ms": t, "units_per_s": K * S * (N - 1) / (t * 1e-3), "distinct_ancestors_per_rank_event": distinct,
            "nodes_materialised_per_sweep": int(sum(distinct)), "logZ_seed0": out['logZ']}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='primate_data')
    ap.add_argument('--K', type=int, default=2048)
    ap.add_argument('--epochs', type=int, default=40)
    ap.add_argument('--lr', type=float, default=0.05)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=24)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'regimes.json'))
    ap.add_argument('--params-out', default=os.path.join(ROOT, 'gpurun_out', 'trained_params.npz'))
    ap.add_argument('--worker', default=None)
    a = ap.parse_args()
    g = load_dataset(a.dataset)['genome']
    N, S, _ = g.shape
    if a.worker:                                         # one configuration per process: the A/B switches are read at phylo_create
        cfg = json.loads(a.worker)
        if cfg.get('flat'):
            Nf, Sf, Kf = cfg['flat']
            lam = np.full(Nf - 1, 10.0)
            print(json.dumps(measure(flat_alignment(Nf, Sf), Kf, M.jc_Q(), M.get_stationary_probs(np.zeros(4) + 0.25), lam, lam, cfg['flags'],
                                     a.reps, Kf <= 4096, jc=True)))
            return
        p = np.load(cfg['params'])
        print(json.dumps(measure(g, a.K, p['Q'], p['pi'], p['lam_l'], p['lam_r'], cfg['flags'], a.reps, True)))
        return
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    init = os.path.join(os.path.dirname(a.params_out), 'untrained_params.npz')
    np.savez(init, Q=M.get_Q(M.init_y_q()), pi=M.get_stationary_probs(np.zeros(4) + 0.25), lam_l=np.full(N - 1, 10.0), lam_r=np.full(N - 1, 10.0))
    v, hist = train(g, a.K, a.epochs, a.lr, a.batch)
    Q, pi, lam_l, lam_r = v.evaluate()
    np.savez(a.params_out, Q=Q, pi=pi, lam_l=lam_l, lam_r=lam_r, elbo_history=np.array(hist))
    print("trained %d epochs (Adam lr %g, batch %d): minibatch logZ %.1f -> %.1f; lam_l %s" % (a.epochs, a.lr, a.batch, hist[0], hist[-1],
                                                                                        np.round(lam_l, 2)), flush=True)
    res = {"dataset": a.dataset, "K": a.K, "N": N, "S": S, "epochs": a.epochs, "elbo_first": hist[0], "elbo_last": hist[-1], "configs": []}
    works = [('primate.p, untrained (lambda = 10, uniform Q)', {'params': init}), ('primate.p, trained', {'params': a.params_out}),
             ('flat 12 x 898 (all-gap rows, JC69), K = 2048', {'flat': [12, 898, 2048]}),
             ('flat 27 x 1949 (DS1 shape, all-gap rows, JC69), K = 4096', {'flat': [27, 1949, 4096]})]
    for pname, wcfg in works:
        for fname, flags, env in (('lazy nodes, launches per rank event', _ffi.FLAGS_DEFAULT, {}),
                                  ('eager nodes, launches per rank event', _ffi.FLAGS_DEFAULT | _ffi.EAGER_NODES, {}),
                                  ('lazy nodes, one launch', _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH, {})):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--dataset', a.dataset, '--K', str(a.K), '--reps', str(a.reps),
                                  '--worker', json.dumps(dict(wcfg, flags=flags))], env=dict(os.environ, **env),
                                 capture_output=True, text=True)
            if out.returncode != 0:
                raise SystemExit(out.stderr[-2000:])
            r = json.loads(out.stdout.strip().splitlines()[-1])
            r.update(parameters=pname, form=fname)
            res["configs"].append(r)
            print("%-58s %-38s t_sweep %.4f ms  %.3e units/s  nodes materialised %5d  distinct ancestors %s"
                  % (pname, fname, r['t_sweep_ms'], r['units_per_s'], r['nodes_materialised_per_sweep'],
                     r['distinct_ancestors_per_rank_event']), flush=True)
    json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
