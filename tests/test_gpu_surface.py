"""GPU tests of the Python surface that mirrors the reference's classes (VCSMC, CSMC, runner)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cpu_ref as O
from phylo_amd import datasets
from phylo_amd.csmc import CSMC, Vertex
from phylo_amd.vcsmc import VCSMC, default_args

pytestmark = pytest.mark.gpu
PI = np.full((1, 4), 0.25)


def test_vcsmc_sample_phylogenies_matches_oracle():
    d = datasets.load_dataset('primate_data_wang')                      # BASELINE config 0: primates_small, JC69, K=16
    v = VCSMC(d, K=16, args=default_args(jcmodel=True, seed=5))
    elbo = v.sample_phylogenies()
    g = d['genome']
    lam = v.left_branches_param            # exp(log 10) = 10.000000000000002, as the reference's tf.exp(Variable)
    ref = CO.sweep(g, O.jc_Q(), PI, lam, lam, 16, 5, jc=True)
    assert elbo == ref['logZ']
    assert np.array_equal(v.log_weights.view(np.uint64), ref['log_weights'].view(np.uint64))
    assert v.log_weights.shape == (8, 16) and v.left_branches.shape == (8, 16)
    # derived attributes of the reference
    assert v.cost == -elbo
    assert v.jump_chain_tensor.shape == (16, 1)
    for s in v.jump_chain_tensor[:, 0]:
        assert sorted(s.split('+')) == sorted(d['taxa'])                # every particle ends as one full tree
    assert (v.v_minus == 9).all()
    assert v.log_likelihood_R.shape == (16,) and np.isfinite(v.log_likelihood_R).all()
    np.testing.assert_allclose(v.compute_log_ZSMC(v.log_weights), elbo, rtol=1e-13)
    # a second call uses the next seed
    assert v.sample_phylogenies() != elbo
    v.close()


def test_vcsmc_ops_surface():
    d = datasets.load_dataset('primate_data')
    g = d['genome']
    v = VCSMC(d, K=6, args=default_args())
    rng = np.random.default_rng(0)
    tl, tr = rng.exponential(0.1, 6), rng.exponential(0.1, 6)
    out = v.broadcast_conditional_likelihood_K(g[:6], g[6:], tl, tr)
    np.testing.assert_allclose(out, O.broadcast_conditional_likelihood_K(v.Qmatrix, g[:6], g[6:], tl, tr), rtol=1e-12)
    one = v.conditional_likelihood(g[0], g[6], tl[0], tr[0])
    np.testing.assert_array_equal(one, out[0])
    core = np.stack([g[:5]] * 6)
    rec = np.ones((6, 5), dtype=np.int32)
    np.testing.assert_allclose(v.compute_forest_posterior(core, rec, 6), O.compute_forest_posterior(PI, core, rec), rtol=1e-13)
    lw = rng.normal(scale=10, size=6)
    c2, r2, jc2, idx = v.resample(core, rec, np.array(d['taxa'][:6]), lw, step=2)
    np.testing.assert_array_equal(idx, O.resample_indices(lw, v.seed, 2))
    np.testing.assert_array_equal(c2, core[idx])
    co, rem, q, jck = v.extend_partial_state(np.array([d['taxa']] * 6, dtype=object), 0)
    assert q == pytest.approx(1 / 66) and jck.shape == (6, 11) and co.shape == (6, 2) and rem.shape == (6, 10)
    v.close()


def test_csmc_surface_known_answers(golden_dir):
    nodes = np.load(os.path.join(golden_dir, "csmc_nodes.npz"))
    d = datasets.load_dataset('load_strings')                           # csmc.py:477 toy strings
    c = CSMC(d)
    v0, v1 = Vertex('S0', d['genome'][0]), Vertex('S1', d['genome'][1])
    ch = Vertex('S0+S1', None)
    ch.left, ch.right, ch.left_branch, ch.right_branch = v0, v1, 2, 2
    ll = c.compute_log_conditional_likelihood(ch)
    assert ll == pytest.approx(float(nodes['known/toy_cherry_2_2']), rel=1e-13)     # -19.67257911375802
    assert ch.data_done and ch.data.shape == (10, 4)
    assert c.compute_log_conditional_likelihood(Vertex('S2', d['genome'][2])) == pytest.approx(10 * np.log(0.25), rel=1e-15)
    # memoised subtree is reused as data (csmc.py:278-298)
    top = Vertex('top', None)
    top.left, top.right, top.left_branch, top.right_branch = ch, Vertex('S2', d['genome'][2]), 0.3, 0.1
    ll_top = c.compute_log_conditional_likelihood(top)
    ref, _ = O.tree_loglik(c.Qmatrix, c.prior, 7, [-1] * 4 + [0, 4], [-1] * 4 + [1, 2], [0] * 4 + [2.0, 0.3],
                           [0] * 4 + [2.0, 0.1], 5, d['genome'])
    assert ll_top == pytest.approx(ref, rel=1e-13)
    np.testing.assert_allclose(c.conditional_likelihood(v0, v1, 2, 2), ch.data, rtol=1e-15)
    np.testing.assert_allclose(c.compute_tree_likelihood(c.prior, ch), np.dot(c.prior, ch.data.T))
    # resample on real-data-sized log-weights does not overflow
    w = np.random.default_rng(0).normal(scale=50, size=(32, 3)) + 800.0
    out = c.resample(w, np.arange(32), 1)
    assert out.shape == (32,) and out.min() >= 0 and out.max() < 32
    c.close()
    # primates_small cherry of SURVEY 8c
    d9 = datasets.load_dataset('primate_data_wang')
    c9 = CSMC(d9)
    a, b = Vertex('a', d9['genome'][0]), Vertex('b', d9['genome'][1])
    ch = Vertex('c', None)
    ch.left, ch.right, ch.left_branch, ch.right_branch = a, b, 0.1, 0.1
    assert c9.compute_log_conditional_likelihood(ch) == pytest.approx(-1335.7912884120358, rel=1e-13)
    c9.close()


def test_csmc_sample_phylogenies_runs():
    d = datasets.load_dataset('load_strings')
    c = CSMC(d)
    lw, probs, norm, root = c.sample_phylogenies(4, resampling=True, showing=False)
    assert lw.shape == (4, 3) and len(probs) == 4 and np.isfinite(norm)
    assert sorted(root.id.split('+')) == sorted(d['taxa'])
    c.close()


def _tree_ids(v):
    out, stack = [], [v]
    while stack:
        x = stack.pop()
        out.append(x.id)
        if x.left is not None:
            stack += [x.left, x.right]
    return sorted(out)


def test_csmc_sample_phylogenies_golden_from_reference(golden_dir):
    """a13 / f4 pinned to the reference itself: tests/golden/csmc_sweeps.npz holds what the reference's
    CSMC.sample_phylogenies(K, resampling=False, showing=False) returned here under random.seed(s) (csmc.py:357-454; with
    resampling off it draws only from Python's `random`, csmc.py:241,392).  Same seed -> same pair picks and same
    random particles for log_likelihood_tilda, so log_weights[K,n-1], the tree-probability table (csmc.py:335-349), norm
    (csmc.py:351-355) and the selected tree must agree; the likelihoods come from phylo_tree_loglik on the GPU."""
    import random
    gold = np.load(os.path.join(golden_dir, "csmc_sweeps.npz"))
    for case in gold['cases']:
        dname, Ktag, stag = str(case).split('/')
        K, seed = int(Ktag[1:]), int(stag[4:])
        d = {'taxa': [str(t) for t in gold['taxa/' + dname]], 'genome': gold['genome/' + dname]}
        c = CSMC(d)
        random.seed(seed)
        lw, probs, norm, root = c.sample_phylogenies(K, resampling=False, showing=False)
        c.close()
        np.testing.assert_allclose(lw, gold[case + '/log_weights'], rtol=1e-10, atol=1e-9, err_msg=str(case))
        # real-data weights overflow in the reference (exp(log w ~ 700) -> inf -> nan/0 probabilities): same here
        np.testing.assert_allclose(np.asarray(probs), gold[case + '/tree_probabilities'], rtol=1e-8, equal_nan=True, err_msg=str(case))
        gn = float(gold[case + '/norm'])
        assert (norm == gn) if not np.isfinite(gn) else norm == pytest.approx(gn, rel=1e-8), (case, norm, gn)
        assert _tree_ids(root) == [str(x) for x in gold[case + '/selected_nodes']], case


def test_runner_end_to_end(capsys, tmp_path, monkeypatch):
    import glob
    import pickle
    import runner
    monkeypatch.chdir(tmp_path)
    elbos = runner.main(['--dataset', 'primate_data_wang', '--n_particles', '16', '--jcmodel', 'true', '--num_epoch', '2'])
    assert len(elbos) == 2 and np.isfinite(elbos).all()
    text = capsys.readouterr().out
    assert 'Initial evaluation of ELBO' in text and 'Epoch 2' in text and 'Done training.' in text
    # the reference's result artefacts (vcsmc.py:503-516, 622-642)
    out = glob.glob(str(tmp_path / 'results' / 'primate_data_wang' / 'False' / '16' / '*'))
    assert len(out) == 1
    assert 'n_particles : 16' in open(out[0] + '/run_parameters.txt').read()
    res = pickle.load(open(out[0] + '/results.p', 'rb'))
    for key in ('cost', 'nParticles', 'nTaxa', 'lr', 'log_weights', 'Qmatrices', 'left_branches', 'right_branches',
                'log_lik', 'll_tilde', 'log_lik_R', 'jump_chain_evolution', 'best_epoch', 'best_log_lik', 'best_jump_chain'):
        assert key in res, key
    assert res['log_weights'].shape == (2, 8, 16) and res['nTaxa'] == 9
    nwk = res['best_newick']
    assert nwk.endswith(';') and nwk.count('(') == 8 and all(('S%d:' % i) in nwk for i in range(9))
    # twisted proposal through the CLI (--twisting is the README's spelling of --nested)
    elbos = runner.main(['--dataset', 'primate_data_wang', '--n_particles', '8', '--twisting', 'true', '--M', '2',
                         '--num_epoch', '1'])
    assert np.isfinite(elbos).all()
