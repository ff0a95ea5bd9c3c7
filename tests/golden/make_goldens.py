#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's NumPy implementation (csmc.py).

Runs only in the build container, where /root/reference exists; the GPU box never sees the
reference.  The outputs are data (inputs + expected outputs); no reference source is copied.

  python tests/golden/make_goldens.py

What is captured:
  * csmc_nodes.npz  - CSMC.conditional_likelihood / compute_log_conditional_likelihood on hand-built
                      Vertex trees (cherries, balanced, caterpillar; 4..12 taxa; with and without gap
                      columns) for csmc's own Q, the JC69 Q, the get_Q-initial Q and random
                      row-softmax Qs: full [S,4] root partials + log-likelihoods.
  * expm_tables.npz - scipy.linalg.expm(Q t) (the call csmc.py:304-305 makes) for t in 1e-6..10.
  * csmc_resample.npz - CSMC.resample (csmc.py:218-228) on toy weights with numpy's global RNG
                      seeded: the uniforms it consumed and the indices it returned.
  * csmc_sweeps.npz - CSMC.sample_phylogenies(K, resampling=False, showing=False) (csmc.py:357-454), the only
                      sweep-level output of the reference that runs here: with resampling off it draws only from
                      Python's `random` (csmc.py:241,392), so random.seed(s) pins it.  Stored per case:
                      log_weights[K,n-1], tree_probabilities[K], norm and the vertex ids of the selected tree
                      (the returned Graph's node names).   `--only sweeps` regenerates this file alone.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

import numpy as np            # noqa: E402
import scipy.linalg as spl    # noqa: E402
import csmc as ref            # noqa: E402  (the reference module)

from phylo_amd.datasets import load_fasta   # noqa: E402

ALPHA = {'A': [1, 0, 0, 0], 'C': [0, 1, 0, 0], 'G': [0, 0, 1, 0], 'T': [0, 0, 0, 1],
         '-': [1, 1, 1, 1], '?': [1, 1, 1, 1]}


def encode(strings):
    g = np.zeros((len(strings), len(strings[0]), 4))
    for i, s in enumerate(strings):
        for j, c in enumerate(s):
            g[i, j] = ALPHA[c]
    return g


def softmax_Q(rng):
    y = rng.normal(size=(4, 4))
    e = np.exp(y)
    np.fill_diagonal(e, 0.0)
    q = e / e.sum(axis=1, keepdims=True)
    np.fill_diagonal(q, -q.sum(axis=1))
    return q


def q_set():
    rng = np.random.default_rng(1234)
    qs = {
        'csmc': np.array([[-1., .25, .5, .25], [.25, -1., .25, .5], [.5, .25, -1., .25],
                          [.25, .5, .25, -1.]]) / 10,
        'jc': np.full((4, 4), 0.25) - np.eye(4),
        'gtr_init': np.full((4, 4), 1.0 / 3) - np.eye(4) * (4.0 / 3),
    }
    for i in range(3):
        qs['rand%d' % i] = softmax_Q(rng)
    return qs


def build_tree(genome, shape, rng):
    """Returns arrays (left, right, bl, br, root) over node ids; leaves 0..L-1."""
    L = genome.shape[0]
    left, right, bl, br = {}, {}, {}, {}
    nxt = L
    if shape == 'caterpillar':
        cur = 0
        for i in range(1, L):
            left[nxt], right[nxt] = cur, i
            cur = nxt
            nxt += 1
        root = cur
    elif shape == 'balanced':
        level = list(range(L))
        while len(level) > 1:
            new = []
            for i in range(0, len(level) - 1, 2):
                left[nxt], right[nxt] = level[i], level[i + 1]
                new.append(nxt)
                nxt += 1
            if len(level) % 2:
                new.append(level[-1])
            level = new
        root = level[0]
    else:
        raise ValueError(shape)
    for i in left:
        bl[i] = float(rng.exponential(0.1) + 1e-3)
        br[i] = float(rng.exponential(0.1) + 1e-3)
    n_nodes = nxt
    la = np.full(n_nodes, -1, dtype=np.int32)
    ra = np.full(n_nodes, -1, dtype=np.int32)
    bla = np.zeros(n_nodes)
    bra = np.zeros(n_nodes)
    for i in left:
        la[i], ra[i], bla[i], bra[i] = left[i], right[i], bl[i], br[i]
    return la, ra, bla, bra, root


def run_ref_tree(Q, genome, la, ra, bla, bra, root):
    L = genome.shape[0]
    c = ref.CSMC({'taxa': ['S%d' % i for i in range(L)], 'genome': genome})
    c.Qmatrix = Q
    verts = [ref.Vertex(id='S%d' % i, data=genome[i]) for i in range(L)]
    for i in range(L, len(la)):
        v = ref.Vertex(id='n%d' % i, data=None)
        verts.append(v)
    for i in range(L, len(la)):
        verts[i].left, verts[i].right = verts[la[i]], verts[ra[i]]
        verts[i].left_branch, verts[i].right_branch = bla[i], bra[i]
    ll = c.compute_log_conditional_likelihood(verts[root])
    return float(ll), np.array(verts[root].data, dtype=np.float64)


def sweep_goldens():
    """csmc.py:357-454 end to end, seeded through Python's `random` (the reference's only RNG with resampling off)."""
    import contextlib
    import io
    import random
    toy = ['ACTTTGAGAG', 'ACTTTGACAG', 'ACTTTGACTG', 'ACTTTGACTC']       # csmc.py:477
    names_small, prim_small = load_fasta(os.path.join(ROOT, 'phylo_amd', 'data', 'primates_small.fa'))
    sets = {'toy': (['S%d' % i for i in range(4)], encode(toy)),
            'primates_small': (list(names_small), encode(prim_small))}
    out, cases = {}, []
    for dname, (taxa, genome) in sets.items():
        for K in (4, 8):
            for seed in (0, 1):
                c = ref.CSMC({'taxa': list(taxa), 'genome': genome})
                random.seed(seed)
                with contextlib.redirect_stdout(io.StringIO()):       # the reference prints its progress
                    lw, probs, norm, G = c.sample_phylogenies(K, resampling=False, showing=False)
                tag = '%s/K%d/seed%d' % (dname, K, seed)
                cases.append(tag)
                out[tag + '/log_weights'] = np.asarray(lw, dtype=np.float64)
                out[tag + '/tree_probabilities'] = np.asarray(probs, dtype=np.float64)
                out[tag + '/norm'] = np.float64(norm)
                out[tag + '/selected_nodes'] = np.array(sorted(G.get_nodes_data()))
    for dname, (taxa, genome) in sets.items():
        out['taxa/' + dname] = np.array(taxa)
        out['genome/' + dname] = genome
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, 'csmc_sweeps.npz'), **out)
    print('wrote', len(cases), 'sample_phylogenies cases')


def main():
    if '--only' in sys.argv and sys.argv[sys.argv.index('--only') + 1] == 'sweeps':
        sweep_goldens()
        return
    out = {}
    qs = q_set()
    rng = np.random.default_rng(20260004)
    toy = ['ACTTTGAGAG', 'ACTTTGACAG', 'ACTTTGACTG', 'ACTTTGACTC']       # csmc.py:477
    _, prim_small = load_fasta(os.path.join(ROOT, 'phylo_amd', 'data', 'primates_small.fa'))
    _, prim = load_fasta(os.path.join(ROOT, 'phylo_amd', 'data', 'primate.fa'))
    datasets = {
        'toy': encode(toy),
        'primates_small': encode(prim_small),
        'primate_gap': encode(prim)[:, 600:898],           # the 30 gap cells live in this window
        'ragged1': encode(['A', 'C', 'G', '-', 'T']),       # S = 1
    }
    cases = []
    for dname, genome in datasets.items():
        for shape in ('caterpillar', 'balanced'):
            la, ra, bla, bra, root = build_tree(genome, shape, rng)
            for qname, Q in qs.items():
                ll, data = run_ref_tree(Q, genome, la, ra, bla, bra, root)
                tag = '%s/%s/%s' % (dname, shape, qname)
                cases.append(tag)
                out[tag + '/left'] = la
                out[tag + '/right'] = ra
                out[tag + '/bl'] = bla
                out[tag + '/br'] = bra
                out[tag + '/root'] = np.int32(root)
                out[tag + '/loglik'] = np.float64(ll)
                out[tag + '/root_data'] = data
    for dname, genome in datasets.items():
        out['genome/' + dname] = genome
    for qname, Q in qs.items():
        out['Q/' + qname] = Q
    out['cases'] = np.array(cases)

    # known answers quoted in SURVEY.md 8c
    c = ref.CSMC({'taxa': ['S%d' % i for i in range(4)], 'genome': datasets['toy']})
    v0, v1 = ref.Vertex('S0', datasets['toy'][0]), ref.Vertex('S1', datasets['toy'][1])
    ch = ref.Vertex('c', None)
    ch.left, ch.right, ch.left_branch, ch.right_branch = v0, v1, 2, 2
    out['known/toy_cherry_2_2'] = np.float64(c.compute_log_conditional_likelihood(ch))
    out['known/toy_leaf2'] = np.float64(
        c.compute_log_conditional_likelihood(ref.Vertex('S2', datasets['toy'][2])))
    c9 = ref.CSMC({'taxa': list(range(9)), 'genome': datasets['primates_small']})
    a, b = ref.Vertex('a', datasets['primates_small'][0]), ref.Vertex('b', datasets['primates_small'][1])
    ch = ref.Vertex('c', None)
    ch.left, ch.right, ch.left_branch, ch.right_branch = a, b, 0.1, 0.1
    out['known/primates_small_cherry_01'] = np.float64(c9.compute_log_conditional_likelihood(ch))
    np.savez_compressed(os.path.join(HERE, 'csmc_nodes.npz'), **out)

    # expm tables: the exact call csmc.py:304-305 makes
    ts = np.array([1e-6, 1e-4, 1e-3, 0.01, 0.05, 0.1, 0.25, 0.5, 0.9, 1.0, 1.3, 2.0, 2.7, 5.0, 7.5,
                   10.0, 25.0, 60.0])
    ex = {'t': ts}
    for qname, Q in qs.items():
        ex['Q/' + qname] = Q
        ex['P/' + qname] = np.stack([spl.expm(Q * t) for t in ts])
    np.savez_compressed(os.path.join(HERE, 'expm_tables.npz'), **ex)

    # CSMC.resample (csmc.py:218-228) with numpy's global RNG seeded
    rs = {}
    for case, (K, seed) in enumerate([(8, 1), (64, 2), (1000, 3)]):
        g = np.random.default_rng(100 + case)
        w = g.normal(scale=3.0, size=(K, 3))
        np.random.seed(seed)
        uniforms = np.random.random_sample(K)          # what np.random.choice will consume
        np.random.seed(seed)
        chain = np.arange(K)
        got = c.resample(w, chain, 1)                  # returns jump_chain_K[indices]
        rs['w%d' % case] = w
        rs['u%d' % case] = uniforms
        rs['idx%d' % case] = np.asarray(got, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'csmc_resample.npz'), **rs)
    print('wrote', len(cases), 'tree cases')
    sweep_goldens()


if __name__ == '__main__':
    main()
