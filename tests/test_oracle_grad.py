"""The gradient oracle (oracle/cpu_grad.py) against central differences of its own forward pass, and the forward
pass against the sweep oracle.  CPU only."""
import numpy as np
import pytest

from oracle import cpu_grad as G
from oracle import cpu_ref as O


def _case(N=6, S=24, K=12, seed=5):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 5, size=(N, S))
    genome = np.zeros((N, S, 4))
    for a in range(4):
        genome[..., a] = (codes == a) | (codes == 4)
    y = rng.normal(size=(4, 4)) * 0.3
    e = np.exp(y)
    np.fill_diagonal(e, 0.0)
    Q = e / e.sum(axis=1, keepdims=True)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    p = np.exp(rng.normal(size=4) * 0.3)
    pi = (p / p.sum())[None, :]
    lam_l = np.exp(rng.normal(size=N - 1) * 0.3 + 1.5)
    lam_r = np.exp(rng.normal(size=N - 1) * 0.3 + 1.5)
    return genome, Q, pi, lam_l, lam_r, K


def test_forward_matches_sweep_oracle():
    genome, Q, pi, ll, lr, K = _case()
    f = G.forward(genome, Q, pi, ll, lr, K, seed=77)
    s = O.sweep(genome, Q, pi, ll, lr, K, 77)
    np.testing.assert_allclose(f['lw'], s['log_weights'], rtol=0, atol=1e-9)
    assert abs(f['logZ'] - s['logZ']) < 1e-9
    f2 = G.forward(genome, Q, pi, ll, lr, K, seed=77, struct=f['struct'])
    assert f2['logZ'] == f['logZ']


@pytest.mark.parametrize("flags", [O.QUIRK_Q1_RAW_Q, 0])
def test_gradient_matches_central_differences(flags):
    genome, Q, pi, ll, lr, K = _case()
    g = G.sweep_grad(genome, Q, pi, ll, lr, K, seed=77, flags=flags)
    st = g['struct']
    for which, key, idxs in (('lam_l', 'd_lam_l', [(0,), (2,), (4,)]), ('lam_r', 'd_lam_r', [(1,), (3,)]),
                             ('pi', 'd_pi', [(0,), (3,)]), ('Q', 'd_Q', [(0, 0), (1, 2), (3, 1)])):
        for idx in idxs:
            fd = G.finite_difference(genome, Q, pi, ll, lr, K, 77, st, which, idx, flags=flags)
            an = g[key][idx]
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (which, idx, fd, an)


def test_chain_rules_to_reference_variables():
    """to_variables against central differences through the reference's parameterisation (vcsmc.py:119-148)."""
    genome, _, _, _, _, K = _case()
    rng = np.random.default_rng(3)
    N = genome.shape[0]
    y_q = rng.normal(size=(4, 4)) * 0.2
    y_s = rng.normal(size=4) * 0.2
    a_l, a_r = rng.normal(size=N - 1) * 0.2 + 1.0, rng.normal(size=N - 1) * 0.2 + 1.0

    def model(y_q, y_s, a_l, a_r):
        e = np.exp(y_q)
        np.fill_diagonal(e, 0.0)
        Q = e / e.sum(axis=1, keepdims=True)
        np.fill_diagonal(Q, -Q.sum(axis=1))
        pi = (np.exp(y_s) / np.exp(y_s).sum())[None, :]
        return Q, pi, np.exp(a_l), np.exp(a_r)

    Q, pi, ll, lr = model(y_q, y_s, a_l, a_r)
    g = G.sweep_grad(genome, Q, pi, ll, lr, K, seed=9)
    v = G.to_variables(Q, pi, ll, lr, g)
    st = g['struct']

    def val(yq, ys, al, ar):
        return G.forward(genome, *model(yq, ys, al, ar), K, 9, struct=st)['logZ']

    h = 1e-6
    for (i, j) in [(0, 1), (2, 0), (3, 2)]:
        d = np.zeros((4, 4)); d[i, j] = h
        fd = (val(y_q + d, y_s, a_l, a_r) - val(y_q - d, y_s, a_l, a_r)) / (2 * h)
        assert abs(fd - v['d_y_q'][i, j]) < 2e-6 * max(1.0, abs(fd))
    assert np.all(np.diag(v['d_y_q']) == 0.0)
    for i in (0, 2):
        d = np.zeros(4); d[i] = h
        fd = (val(y_q, y_s + d, a_l, a_r) - val(y_q, y_s - d, a_l, a_r)) / (2 * h)
        assert abs(fd - v['d_y_station'][i]) < 2e-6 * max(1.0, abs(fd))
    for i in (0, 3):
        d = np.zeros(N - 1); d[i] = h
        fd = (val(y_q, y_s, a_l + d, a_r) - val(y_q, y_s, a_l - d, a_r)) / (2 * h)
        assert abs(fd - v['d_loglam_l'][i]) < 2e-6 * max(1.0, abs(fd))
        fd = (val(y_q, y_s, a_l, a_r + d) - val(y_q, y_s, a_l, a_r - d)) / (2 * h)
        assert abs(fd - v['d_loglam_r'][i]) < 2e-6 * max(1.0, abs(fd))


def test_twisted_forward_matches_sweep_oracle():
    genome, Q, pi, ll, lr, K = _case(N=5, S=12, K=8)
    f = G.forward_twisted(genome, Q, pi, ll, lr, K, 2, seed=31)
    s = O.sweep_twisted(genome, Q, pi, ll, lr, K, 2, 31)
    np.testing.assert_allclose(f['lw'], s['log_weights'], rtol=0, atol=1e-9)
    assert abs(f['logZ'] - s['logZ']) < 1e-9
    assert np.array_equal(np.stack(f['co']), s['merges'])
    f2 = G.forward_twisted(genome, Q, pi, ll, lr, K, 2, seed=31, struct=f['struct'])
    assert f2['logZ'] == f['logZ']


@pytest.mark.parametrize("M", [1, 3])
def test_twisted_gradient_matches_central_differences(M):
    """vncsmc.py:295-416: the potentials of every (pair, sub-sample) are differentiated, the draws are not."""
    genome, Q, pi, ll, lr, K = _case(N=5, S=12, K=8)
    g = G.sweep_grad_twisted(genome, Q, pi, ll, lr, K, M, seed=31)
    st = g['struct']
    for which, key, idxs in (('lam_l', 'd_lam_l', [(0,), (2,), (3,)]), ('lam_r', 'd_lam_r', [(1,), (3,)]),
                             ('pi', 'd_pi', [(0,), (3,)]), ('Q', 'd_Q', [(0, 0), (1, 2), (3, 1)])):
        for idx in idxs:
            fd = G.finite_difference_twisted(genome, Q, pi, ll, lr, K, M, 31, st, which, idx)
            an = g[key][idx]
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (which, idx, fd, an)
