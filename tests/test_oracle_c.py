"""Pin the C oracle (oracle/csrc/oracle.c, the bit-exact mirror of the device arithmetic) to the NumPy
oracle, to the csmc.py golden vectors and to scipy/numpy for exp, log, expm."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cpu_ref as O

PI = np.full((1, 4), 0.25)


def ulp_err(got, ref):
    return np.max(np.abs(got - ref) / np.spacing(np.abs(ref)))


def test_exp_log_within_one_ulp_of_libm():
    rng = np.random.default_rng(0)
    x = rng.uniform(-740, 700, 300000)
    assert ulp_err(CO.math_probe(0, x), np.exp(x)) <= 1.0
    x = np.exp(rng.uniform(-700, 700, 300000))
    assert ulp_err(CO.math_probe(1, x), np.log(x)) <= 1.0
    x = rng.uniform(0.6, 1.6, 300000)
    assert ulp_err(CO.math_probe(1, x), np.log(x)) <= 1.0
    # special values
    sp = CO.math_probe(1, np.array([0.0, 1.0, np.inf, -1.0, 5e-324]))
    assert sp[0] == -np.inf and sp[1] == 0.0 and sp[2] == np.inf and np.isnan(sp[3])
    assert sp[4] == pytest.approx(np.log(5e-324), rel=1e-15)
    se = CO.math_probe(0, np.array([0.0, -800.0, 800.0, -745.0, 1e-10]))
    assert se[0] == 1.0 and se[1] == 0.0 and se[2] == np.inf
    assert se[3] == pytest.approx(np.exp(-745.0), rel=0.5) and se[4] == pytest.approx(np.exp(1e-10), rel=1e-15)


def test_philox_matches_numpy_oracle_and_kat():
    assert CO.philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    x = O.philox4x32(123, 7, 2, 5, 0xdeadbeefcafef00d)
    assert CO.philox(123, 7, 2, 5, 0xdeadbeefcafef00d) == [int(v) for v in x]


def test_expm_against_scipy_tables(golden_dir):
    ex = np.load(os.path.join(golden_dir, "expm_tables.npz"))
    for q in ('csmc', 'jc', 'gtr_init', 'rand0', 'rand1', 'rand2'):
        np.testing.assert_allclose(CO.expm_batched(ex['Q/' + q], ex['t']), ex['P/' + q], rtol=0, atol=1e-14)
    np.testing.assert_allclose(CO.expm_batched(ex['Q/jc'], ex['t'], jc=True), ex['P/jc'], rtol=0, atol=3e-15)
    # all five Pade orders and the squaring branch are exercised: norm thresholds / ||Q||_1 = 2
    t = np.array([1e-3, 0.05, 0.3, 0.9, 2.5, 40.0])
    Q = ex['Q/gtr_init']
    np.testing.assert_allclose(CO.expm_batched(Q, t), O.expm_batched(Q, t), rtol=0, atol=1e-14)


def test_tree_cases_match_reference_goldens(golden_dir):
    nodes = np.load(os.path.join(golden_dir, "csmc_nodes.npz"))
    prior = np.ones(4) / 4
    for tag in nodes['cases']:
        dname, shape, qname = str(tag).split('/')
        ll, data = CO.tree_loglik(nodes['Q/' + qname], prior, nodes[tag + '/left'], nodes[tag + '/right'],
                                  nodes[tag + '/bl'], nodes[tag + '/br'], int(nodes[tag + '/root']),
                                  nodes['genome/' + dname])
        np.testing.assert_allclose(data, nodes[tag + '/root_data'], rtol=1e-12, atol=0)
        assert ll == pytest.approx(float(nodes[tag + '/loglik']), rel=1e-12)
    assert CO.tree_loglik(nodes['Q/csmc'], prior, [-1] * 4 + [0], [-1] * 4 + [1], [0] * 4 + [2.0], [0] * 4 + [2.0], 4,
                          nodes['genome/toy'])[0] == pytest.approx(-19.67257911375802, rel=1e-13)


def test_ops_match_numpy_oracle():
    rng = np.random.default_rng(1)
    g = O.form_dataset_from_strings(['ACGT-ACGTTGCA?', 'ACGTTACGTAGCAA', 'TCGT-ACGATGCAG'], O.ALPHABET_DIR_BLANK)['genome']
    Q = O.get_Q(rng.normal(size=(4, 4)))
    K = 5
    l, r = g[rng.integers(0, 3, K)], g[rng.integers(0, 3, K)] * 0.5
    tl, tr = rng.exponential(0.1, K), rng.exponential(0.1, K)
    np.testing.assert_allclose(CO.cond_likelihood_K(Q, l, r, tl, tr), O.broadcast_conditional_likelihood_K(Q, l, r, tl, tr),
                               rtol=1e-13)
    core = g[rng.integers(0, 3, (K, 2))] * rng.uniform(0.1, 1, (K, 2, 14, 1))
    rec = rng.integers(1, 4, (K, 2)).astype(np.int32)
    np.testing.assert_allclose(CO.forest_loglik(PI, core, rec), O.compute_forest_posterior(PI, core, rec), rtol=1e-13)
    lw = rng.normal(scale=25, size=(3, 500)) - 3000
    assert CO.log_zsmc(lw) == pytest.approx(O.compute_log_ZSMC(lw), rel=1e-13)
    for step in (1, 2):
        np.testing.assert_array_equal(CO.resample(lw[0], 9, step), O.resample_indices(lw[0], 9, step))


@pytest.mark.parametrize("jc", [True, False])
def test_sweep_matches_numpy_oracle(jc):
    g = O.form_dataset_from_strings(['ACTTTGAGAGAC', 'ACTTTGACAGTT', 'ACTTTGACTG-A', 'ACTTTGACTCAA', 'AC-TTGACTCGG',
                                     'GCTTAGACTCGA'], O.ALPHABET_DIR_BLANK)['genome']
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    for K, seed in [(8, 0), (100, 4)]:
        a = O.sweep(g, Q, PI, lam, lam, K, seed)
        b = CO.sweep(g, Q, PI, lam, lam, K, seed, jc=jc)
        np.testing.assert_array_equal(a['ancestors'], b['ancestors'])
        np.testing.assert_array_equal(a['merges'], b['merges'])
        np.testing.assert_allclose(a['log_weights'], b['log_weights'], rtol=1e-11)
        np.testing.assert_allclose(a['left_branches'], b['left_branches'], rtol=1e-14)
        assert a['logZ'] == pytest.approx(b['logZ'], rel=1e-12)


@pytest.mark.parametrize("coded", [True, False])
def test_twisted_sweep_matches_numpy_oracle(coded):
    """The C oracle prices the look-ahead potentials of coded leaves by code (contracts v3 / v4: leaf-leaf pairs by
    code pair, leaf x internal pairs by leaf code); the NumPy oracle sums the reference's formula over sites.  Same
    trajectories, weights within 1e-9 -- on coded data (both regroupings in use) and on generic data (neither)."""
    if coded:
        g = O.form_dataset_from_strings(['ACTTTGAGAGAC', 'ACTTTGACAGTT', 'ACTTTGACTG-A', 'ACTTTGACTCAA', 'AC-TTGACTCGG',
                                         'GCTTAGACTCGA'], O.ALPHABET_DIR_BLANK)['genome']
    else:
        g = np.random.default_rng(8).uniform(0.05, 1.0, size=(6, 12, 4))
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    for K, M, seed in [(6, 1, 0), (10, 3, 5)]:
        a = O.sweep_twisted(g, Q, PI, lam, lam, K, M, seed)
        b = CO.sweep_twisted(g, Q, PI, lam, lam, K, M, seed)
        np.testing.assert_array_equal(a['ancestors'], b['ancestors'])
        np.testing.assert_array_equal(a['merges'], b['merges'])
        np.testing.assert_allclose(a['log_weights'], b['log_weights'], rtol=1e-9)
        assert a['logZ'] == pytest.approx(b['logZ'], rel=1e-10)


def test_sweep_is_thread_count_invariant():
    g = O.form_dataset_from_strings(['ACTTTGAGAG', 'ACTTTGACAG', 'ACTTTGACTG', 'ACTTTGACTC'], O.ALPHABET_DIR)['genome']
    lam = np.full(3, 10.0)
    n = CO.num_threads()
    CO.set_threads(1)
    a = CO.sweep(g, O.jc_Q(), PI, lam, lam, 64, 3, jc=True)
    CO.set_threads(max(n, 2))
    b = CO.sweep(g, O.jc_Q(), PI, lam, lam, 64, 3, jc=True)
    assert np.array_equal(a['log_weights'].view(np.uint64), b['log_weights'].view(np.uint64))
    assert a['logZ'] == b['logZ']
