"""Pin the NumPy oracle (oracle/cpu_ref.py) to vectors produced by the reference's csmc.py
(tests/golden/make_goldens.py) and to the known-answer identities of SURVEY.md section 4."""
import os

import numpy as np
import pytest

from oracle import cpu_ref as O

RTOL = 1e-12   # fp64; scipy expm both sides, only summation order differs


@pytest.fixture(scope="module")
def nodes(golden_dir):
    return np.load(os.path.join(golden_dir, "csmc_nodes.npz"))


def test_known_answers_from_survey(nodes):
    # SURVEY 8c values, regenerated from csmc.py and stored in the fixture
    assert float(nodes['known/toy_cherry_2_2']) == pytest.approx(-19.67257911375802, rel=1e-14)
    assert float(nodes['known/toy_leaf2']) == pytest.approx(-13.862943611198906, rel=1e-14)
    assert float(nodes['known/primates_small_cherry_01']) == pytest.approx(-1335.7912884120358, rel=1e-14)
    Q = nodes['Q/csmc']
    g = nodes['genome/toy']
    prior = np.ones(4) / 4
    ll, _ = O.tree_loglik(Q, prior, 5, [-1] * 4 + [0], [-1] * 4 + [1], [0] * 4 + [2.0], [0] * 4 + [2.0], 4, g)
    assert ll == pytest.approx(float(nodes['known/toy_cherry_2_2']), rel=RTOL)
    ll, _ = O.tree_loglik(Q, prior, 4, [-1] * 4, [-1] * 4, [0] * 4, [0] * 4, 2, g)
    assert ll == pytest.approx(10 * np.log(0.25), rel=1e-15)


def test_tree_cases_match_reference(nodes):
    prior = np.ones(4) / 4
    for tag in nodes['cases']:
        dname, shape, qname = str(tag).split('/')
        g = nodes['genome/' + dname]
        Q = nodes['Q/' + qname]
        la, ra = nodes[tag + '/left'], nodes[tag + '/right']
        ll, data = O.tree_loglik(Q, prior, len(la), la, ra, nodes[tag + '/bl'], nodes[tag + '/br'],
                                 int(nodes[tag + '/root']), g)
        np.testing.assert_allclose(data, nodes[tag + '/root_data'], rtol=RTOL, atol=0)
        assert ll == pytest.approx(float(nodes[tag + '/loglik']), rel=RTOL)


def test_broadcast_K_equals_per_particle_reference_formula(nodes):
    g = nodes['genome/primates_small']
    Q = nodes['Q/rand1']
    rng = np.random.default_rng(5)
    K = 6
    li, ri = rng.integers(0, 9, K), rng.integers(0, 9, K)
    tl, tr = rng.exponential(0.1, K), rng.exponential(0.1, K)
    out = O.broadcast_conditional_likelihood_K(Q, g[li], g[ri], tl, tr)
    for k in range(K):
        ref = O.conditional_likelihood(Q, g[li[k]], g[ri[k]], tl[k], tr[k])
        np.testing.assert_allclose(out[k], ref, rtol=1e-14)


def test_expm_tables_and_closed_forms(golden_dir):
    ex = np.load(os.path.join(golden_dir, "expm_tables.npz"))
    t = ex['t']
    np.testing.assert_allclose(O.jc69_closed_form(t), ex['P/jc'], rtol=0, atol=2e-15)
    # jcmodel=false at initialisation: P_ii = 1/4 + 3/4 exp(-4t/3)
    P = ex['P/gtr_init']
    np.testing.assert_allclose(P[:, 0, 0], 0.25 + 0.75 * np.exp(-4 * t / 3), atol=3e-15)
    for q in ('csmc', 'jc', 'gtr_init', 'rand0', 'rand1', 'rand2'):
        np.testing.assert_allclose(ex['P/' + q].sum(axis=2), 1.0, atol=1e-13)
        np.testing.assert_allclose(O.expm_batched(ex['Q/' + q], t), ex['P/' + q], rtol=0, atol=1e-15)


def test_model_matrices():
    Q = O.get_Q(O.init_y_q())
    np.testing.assert_allclose(Q, np.full((4, 4), 1 / 3) - np.eye(4) * 4 / 3, atol=1e-15)
    np.testing.assert_allclose(O.jc_Q(), np.full((4, 4), 0.25) - np.eye(4), atol=0)
    np.testing.assert_allclose(O.get_stationary_probs(np.zeros(4) + 0.25), np.full((1, 4), 0.25), atol=1e-16)


def test_log_double_factorial_identities():
    # 0 for n in {1,2} leaves, log 3 for n=3, log 15 for n=4
    n = np.array([1, 2, 3, 4])
    got = O.log_double_factorial(2 * np.maximum(n, 2) - 3)
    np.testing.assert_allclose(got, [0.0, 0.0, np.log(3.0), np.log(15.0)], rtol=1e-15)


def test_csmc_resample_matches_reference(golden_dir):
    rs = np.load(os.path.join(golden_dir, "csmc_resample.npz"))
    for case in range(3):
        idx = O.resample_csmc(rs['w%d' % case], rs['u%d' % case], 1)
        np.testing.assert_array_equal(idx, rs['idx%d' % case])


def test_gap_and_leaf_identities():
    # a gap column [1,1,1,1] contributes log 1 = 0 to a leaf; a non-gap leaf S log 1/4
    g = O.form_dataset_from_strings(['AC-T?'], O.ALPHABET_DIR_BLANK)['genome']
    pi = np.full((1, 4), 0.25)
    ll = O.compute_forest_posterior(pi, g[None], np.ones((1, 1), dtype=np.int32))
    assert ll[0] == pytest.approx(3 * np.log(0.25), rel=1e-15)
    with pytest.raises(KeyError):
        O.form_dataset_from_strings(['ACN'], O.ALPHABET_DIR_BLANK)


def test_sweep_row0_and_structure():
    d = O.form_dataset_from_strings(['ACTTTGAGAG', 'ACTTTGACAG', 'ACTTTGACTG', 'ACTTTGACTC', 'AC-TTGACTC'],
                                    O.ALPHABET_DIR_BLANK)
    N = 5
    K = 32
    out = O.sweep(d['genome'], O.jc_Q(), np.full((1, 4), 0.25), np.full(N - 1, 10.0), np.full(N - 1, 10.0),
                  K, seed=7)
    assert out['log_weights'].shape == (N - 1, K)
    assert out['ancestors'].shape == (N - 2, K)
    assert np.all(out['final_record'] == N)
    assert np.isfinite(out['logZ'])
    # row 0 of log_weights contributes exactly 0
    assert O.compute_log_ZSMC(np.zeros((1, K))) == pytest.approx(0.0, abs=1e-15)
    # determinism under a fixed seed; a different seed changes the draws
    out2 = O.sweep(d['genome'], O.jc_Q(), np.full((1, 4), 0.25), np.full(N - 1, 10.0), np.full(N - 1, 10.0),
                   K, seed=7)
    np.testing.assert_array_equal(out['ancestors'], out2['ancestors'])
    assert out['logZ'] == out2['logZ']
    out3 = O.sweep(d['genome'], O.jc_Q(), np.full((1, 4), 0.25), np.full(N - 1, 10.0), np.full(N - 1, 10.0),
                   K, seed=8)
    assert out3['logZ'] != out['logZ']


def test_philox_known_answer():
    # Random123 KAT for philox4x32-10: counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    x = O.philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in x] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    # counter = key = ffffffff... -> 408f276d 41c83b0e a20bc7c6 6d5451fd
    x = O.philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffffffffffff)
    assert [int(v) for v in x] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
