"""GPU parity of the reverse pass (phylo_sweep_backward) against the gradient oracle (oracle/cpu_grad.py), which
tests/test_oracle_grad.py checks against central differences.  Tolerance: relative 1e-9 of the largest entry of
each gradient block (floating point, different summation orders; the forward sweep stays bit-exact)."""
import numpy as np
import pytest

from oracle import cpu_grad as G
from oracle import cpu_ref as O
from phylo_amd import _ffi
from phylo_amd.datasets import load_dataset

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _model(rng, N, spread=0.3, lam=2.0):
    y = rng.normal(size=(4, 4)) * spread
    e = np.exp(y)
    np.fill_diagonal(e, 0.0)
    Q = e / e.sum(axis=1, keepdims=True)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    p = np.exp(rng.normal(size=4) * spread)
    pi = (p / p.sum())[None, :]
    return Q, pi, np.exp(rng.normal(size=N - 1) * spread + lam), np.exp(rng.normal(size=N - 1) * spread + lam)


def _codes_genome(rng, N, S):
    codes = rng.integers(0, 5, size=(N, S))
    g = np.zeros((N, S, 4))
    for a in range(4):
        g[..., a] = (codes == a) | (codes == 4)
    return g


def _check(genome, Q, pi, ll, lr, K, seed, flags=_ffi.FLAGS_DEFAULT, jc=False):
    N, S, _ = genome.shape
    with _ffi.Context(K, N, S) as ctx:
        ctx.set_leaves(genome)
        ctx.set_model(Q, pi, ll, lr, jc69_closed_form=jc)
        plain = ctx.sweep(seed, flags)
        out = ctx.sweep(seed, flags | _ffi.KEEP_GRAPH)
        # keeping the graph does not change the sweep
        for key in ('log_weights', 'log_likelihood', 'ancestors', 'merges'):
            assert np.array_equal(plain[key], out[key]), key
        assert plain['logZ'] == out['logZ']
        g = ctx.sweep_backward()
        g2 = ctx.sweep_backward()                       # deterministic, and repeatable on the kept graph
        for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
            assert np.array_equal(g[key], g2[key]), key
    # oracle on the device's discrete structure (only the ancestors depend on floating point)
    f = G.forward(genome, Q, pi, ll, lr, K, seed, flags)
    st = f['struct']
    for r in range(1, N - 1):
        st['anc'][r] = out['ancestors'][r - 1].astype(np.int64)
    ref = G.sweep_grad(genome, Q, pi, ll, lr, K, seed, flags, struct=st)
    assert abs(ref['logZ'] - out['logZ']) < 1e-9 * max(1.0, abs(out['logZ']))
    for key in ('d_lam_l', 'd_lam_r') + (() if jc else ('d_pi', 'd_Q')):
        scale = max(np.max(np.abs(ref[key])), 1e-300)
        err = np.max(np.abs(g[key] - ref[key])) / scale
        assert err < RTOL, (key, err, g[key], ref[key])
    return g, ref


def test_gradient_small_random_model():
    rng = np.random.default_rng(11)
    genome = _codes_genome(rng, 6, 24)
    Q, pi, ll, lr = _model(rng, 6)
    _check(genome, Q, pi, ll, lr, K=12, seed=77)


def test_gradient_log_q_form_and_two_tiles():
    rng = np.random.default_rng(12)
    genome = _codes_genome(rng, 7, 300)                  # two site tiles in pg_nodes, ragged last tile
    Q, pi, ll, lr = _model(rng, 7)
    _check(genome, Q, pi, ll, lr, K=48, seed=5, flags=0)


def test_gradient_generic_leaves():
    """Leaves that are neither one-hot nor all-ones (no leaf codes)."""
    rng = np.random.default_rng(13)
    genome = rng.uniform(0.05, 1.0, size=(5, 40, 4))
    Q, pi, ll, lr = _model(rng, 5)
    _check(genome, Q, pi, ll, lr, K=16, seed=3)


def test_gradient_primate_subset_initial_model():
    """The reference's initial model (uniform y_q / y_station, rate e^branch_prior) on real sites."""
    genome = load_dataset('primate_data')['genome'][:8, :200]
    N = genome.shape[0]
    Q = np.full((4, 4), 1.0 / 3.0)
    np.fill_diagonal(Q, -1.0)
    pi = np.full((1, 4), 0.25)
    lam = np.full(N - 1, np.exp(np.log(10.0)))
    _check(genome, Q, pi, lam, lam, K=64, seed=2024)


def test_gradient_jc69_rates_only():
    genome = load_dataset('primate_data_wang')['genome'][:, :64]
    N = genome.shape[0]
    lam = np.full(N - 1, 10.0)
    _check(genome, O.jc_Q(), np.full((1, 4), 0.25), lam, lam, K=32, seed=8, jc=True)


def test_backward_needs_graph():
    genome = load_dataset('primate_data_wang')['genome'][:, :32]
    N = genome.shape[0]
    with _ffi.Context(8, N, 32) as ctx:
        ctx.set_leaves(genome)
        ctx.set_model(O.jc_Q(), np.full((1, 4), 0.25), np.full(N - 1, 10.0), np.full(N - 1, 10.0))
        ctx.sweep(1)
        with pytest.raises(_ffi.PhyloError):
            ctx.sweep_backward()
        with pytest.raises(_ffi.PhyloError):
            ctx.sweep(1, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH | _ffi.TWISTING)
