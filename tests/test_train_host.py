"""Host side of the VI training step (phylo_amd/train.py): parameterisation, chain rules, optimiser update rules.
CPU only; the device gradient itself is covered by tests/test_gpu_grad.py."""
import numpy as np
import pytest

from oracle import cpu_grad as G
from phylo_amd import model
from phylo_amd import train as T


def test_variables_initial_values_match_reference_init():
    v = T.Variables(12, np.log(10), jcmodel=False)
    Q, pi, ll, lr = v.evaluate()
    assert np.allclose(np.diag(Q), -1.0) and np.allclose(Q[0, 1], 1 / 3)       # uniform row-softmax
    assert np.allclose(pi, 0.25) and pi.shape == (1, 4)
    assert np.array_equal(ll, model.branch_rates(12, np.log(10))) and np.array_equal(ll, lr)
    assert v.names() == ('a_l', 'a_r', 'y_q', 'y_station')
    vj = T.Variables(5, 1.0, jcmodel=True)
    assert vj.names() == ('a_l', 'a_r')
    assert np.array_equal(vj.evaluate()[0], model.jc_Q())


def test_chain_rules_match_oracle_statement():
    rng = np.random.default_rng(0)
    v = T.Variables(7, 1.0, jcmodel=False)
    v.y_q = rng.normal(size=(4, 4)) * 0.3
    np.fill_diagonal(v.y_q, 0.0)
    v.y_station = rng.normal(size=4) * 0.3
    v.a_l, v.a_r = rng.normal(size=6), rng.normal(size=6)
    Q, pi, ll, lr = v.evaluate()
    raw = {'d_lam_l': rng.normal(size=6), 'd_lam_r': rng.normal(size=6), 'd_pi': rng.normal(size=4), 'd_Q': rng.normal(size=(4, 4))}
    mine = T.chain_rules(v, Q, pi, ll, lr, raw)
    ref = G.to_variables(Q, pi, ll, lr, raw)
    np.testing.assert_allclose(mine['a_l'], ref['d_loglam_l'], rtol=1e-14)
    np.testing.assert_allclose(mine['a_r'], ref['d_loglam_r'], rtol=1e-14)
    np.testing.assert_allclose(mine['y_station'], ref['d_y_station'], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(mine['y_q'], ref['d_y_q'], rtol=1e-13, atol=1e-15)
    assert np.all(np.diag(mine['y_q']) == 0.0)


def test_gradient_descent_moves_up_the_elbo_gradient():
    v = T.Variables(4, 0.5, jcmodel=True)
    g = {'a_l': np.array([1.0, -2.0, 0.5]), 'a_r': np.array([0.0, 4.0, -1.0])}
    T.GradientDescent(0.1).apply(v, g)                       # minimises cost = -logZ
    np.testing.assert_allclose(v.a_l, 0.5 + 0.1 * g['a_l'])
    np.testing.assert_allclose(v.a_r, 0.5 + 0.1 * g['a_r'])


def test_adam_update_rule_tf1_defaults():
    v = T.Variables(3, 0.0, jcmodel=True)
    opt = T.Adam(0.01)
    g1 = {'a_l': np.array([2.0, -1.0]), 'a_r': np.array([0.5, 0.0])}
    opt.apply(v, g1)
    # first step of Adam: m/(sqrt(v)) = sign(g) up to epsilon, step size lr
    np.testing.assert_allclose(v.a_l, [0.01, -0.01], rtol=1e-6)
    np.testing.assert_allclose(v.a_r, [0.01, 0.0], rtol=1e-6, atol=1e-12)
    g2 = {'a_l': np.array([1.0, -1.0]), 'a_r': np.array([0.5, 0.0])}
    opt.apply(v, g2)
    m = 0.9 * (0.1 * -2.0) + 0.1 * -1.0
    s = 0.999 * (0.001 * 4.0) + 0.001 * 1.0
    lr_t = 0.01 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    np.testing.assert_allclose(v.a_l[0], 0.01 - lr_t * m / (np.sqrt(s) + 1e-8), rtol=1e-6)
    assert str(opt).startswith('AdamOptimizer') and str(T.make_optimizer('x', 0.1)).startswith('GradientDescent')
    assert isinstance(T.make_optimizer('Adam', 0.1), T.Adam)


def test_batch_slices_partition_and_rng_sequence():
    """vcsmc.py:453-464: S // b slices of b sites drawn without replacement from the unused sites (python's global RNG),
    then the leftover; with b >= S there is only the leftover slice, so no training step runs (quirk Q9)."""
    import random
    from phylo_amd.vcsmc import VCSMC
    S, b = 898, 256
    data = np.zeros((1, 12, S, 4))
    random.seed(3)
    slices = VCSMC.batch_slices(None, data, b)
    assert [len(s) for s in slices] == [256, 256, 256, 130]
    assert sorted(sum(slices, [])) == list(range(S))                       # a partition of the sites
    random.seed(3)
    assert slices[0] == random.sample(list(range(S)), b)                   # first draw: straight from the RNG stream
    rest = list(set(range(S)) - set(slices[0]))
    assert slices[1] == random.sample(rest, b)                             # second draw: over the set-difference order
    assert [len(s) for s in VCSMC.batch_slices(None, data, 1000)] == [898]
    assert [len(s) for s in VCSMC.batch_slices(None, np.zeros((1, 2, 512, 4)), 256)] == [256, 256]


@pytest.mark.parametrize("jc", [False, True])
@pytest.mark.parametrize("name", ['Adam', 'GradientDescentOptimizer'])
def test_library_optimisers_equal_the_numpy_ones_bit_for_bit(jc, name):
    """phylo_vi_apply (phylo_train.h; no GPU needed) against GradientDescent / Adam of phylo_amd/train.py over five steps."""
    rng = np.random.default_rng(5)
    v1, v2 = T.Variables(7, np.log(10.0), jc), T.Variables(7, np.log(10.0), jc)
    o1, o2 = T.make_optimizer(name, 0.05), T.make_optimizer(name, 0.05)
    for _ in range(5):
        g = {'a_l': rng.normal(size=6), 'a_r': rng.normal(size=6)}
        if not jc:
            g['y_q'], g['y_station'] = rng.normal(size=(4, 4)), rng.normal(size=4)
        packed = np.concatenate([g['a_l'], g['a_r'], g.get('y_q', np.zeros((4, 4))).reshape(-1), g.get('y_station', np.zeros(4))])
        o1.apply(v1, g)
        o2.apply_packed(v2, packed)
    for n in ('a_l', 'a_r', 'y_q', 'y_station'):
        assert np.array_equal(np.asarray(getattr(v1, n)).view(np.uint64), np.asarray(getattr(v2, n)).view(np.uint64)), n
