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
        ctx.sweep(1, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
        ctx.set_model(O.jc_Q(), np.full((1, 4), 0.25), np.full(N - 1, 9.0), np.full(N - 1, 9.0))
        with pytest.raises(_ffi.PhyloError):             # the kept graph belongs to the model it was swept with
            ctx.sweep_backward()


# ---- the twisted proposal's reverse pass (vncsmc.py:295-416) -----------------------------------------------------
def _check_twisted(genome, Q, pi, ll, lr, K, M, seed, jc=False):
    N, S, _ = genome.shape
    flags = _ffi.FLAGS_DEFAULT | _ffi.TWISTING
    with _ffi.Context(K, N, S) as ctx:
        ctx.set_leaves(genome)
        ctx.set_model(Q, pi, ll, lr, jc69_closed_form=jc)
        plain = ctx.sweep(seed, flags, M)
        out = ctx.sweep(seed, flags | _ffi.KEEP_GRAPH, M)
        for key in ('log_weights', 'log_likelihood', 'ancestors', 'merges', 'left_branches', 'right_branches'):
            assert np.array_equal(plain[key], out[key]), key
        assert plain['logZ'] == out['logZ']
        g = ctx.sweep_backward()
        g2 = ctx.sweep_backward()
        for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
            assert np.array_equal(g[key], g2[key]), key
        # a plain keep-graph sweep on the same context afterwards still gives the plain gradient
        ctx.sweep(seed, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
        gp = ctx.sweep_backward()
    # oracle on the device's discrete structure: ancestors, chosen pair (merges) and chosen sub-sample (the branch length)
    f = G.forward_twisted(genome, Q, pi, ll, lr, K, M, seed)
    st = f['struct']
    for r in range(N - 1):
        if r > 0:
            st['anc'][r] = out['ancestors'][r - 1].astype(np.int64)
        pairs = O.pair_list(N - r)
        b_l = -np.log(st['Ul'][r]) / ll[r]
        js = np.zeros(K, dtype=np.int64)
        for k in range(K):
            t = pairs.index((int(out['merges'][r, k, 0]), int(out['merges'][r, k, 1])))
            js[k] = t * M + int(np.argmin(np.abs(b_l[k, t * M:(t + 1) * M] - out['left_branches'][r, k])))
        st['js'][r] = js
    ref = G.sweep_grad_twisted(genome, Q, pi, ll, lr, K, M, seed, struct=st)
    assert abs(ref['logZ'] - out['logZ']) < 1e-9 * max(1.0, abs(out['logZ']))
    for key in ('d_lam_l', 'd_lam_r') + (() if jc else ('d_pi', 'd_Q')):
        scale = max(np.max(np.abs(ref[key])), 1e-300)
        err = np.max(np.abs(g[key] - ref[key])) / scale
        assert err < RTOL, (key, err, g[key], ref[key])
    rp = G.sweep_grad(genome, Q, pi, ll, lr, K, seed)
    assert np.max(np.abs(gp['d_lam_l'] - rp['d_lam_l'])) < 1e-6 * max(1.0, np.max(np.abs(rp['d_lam_l'])))
    return g, ref


@pytest.mark.parametrize("M", [1, 3])
def test_twisted_gradient_small_random_model(M):
    rng = np.random.default_rng(31)
    genome = _codes_genome(rng, 6, 24)
    Q, pi, ll, lr = _model(rng, 6)
    _check_twisted(genome, Q, pi, ll, lr, K=12, M=M, seed=77)


def test_twisted_gradient_two_site_groups_generic_leaves():
    rng = np.random.default_rng(32)
    genome = rng.uniform(0.05, 1.0, size=(5, 300, 4))      # two groups of 256 sites in pg_twist_xchunks, ragged
    Q, pi, ll, lr = _model(rng, 5)
    _check_twisted(genome, Q, pi, ll, lr, K=40, M=2, seed=5)   # 40 adopters: nodes with more than PG_XCH entries


def test_twisted_gradient_jc69_rates_only():
    genome = load_dataset('primate_data_wang')['genome'][:6, :64]
    N = genome.shape[0]
    lam = np.full(N - 1, 10.0)
    _check_twisted(genome, O.jc_Q(), np.full((1, 4), 0.25), lam, lam, K=16, M=2, seed=8, jc=True)


# ---- the training step built on the reverse pass (phylo_amd/train.py, VCSMC.train) ----------------------------
def test_trainer_gradients_match_oracle_chain_rules():
    from phylo_amd import train as T
    rng = np.random.default_rng(21)
    genome = _codes_genome(rng, 6, 96)
    v = T.Variables(6, 1.2, jcmodel=False)
    v.y_q = rng.normal(size=(4, 4)) * 0.2
    np.fill_diagonal(v.y_q, 0.0)
    v.y_station = rng.normal(size=4) * 0.2
    sites = list(rng.permutation(96)[:32])
    tr = T.Trainer(genome, 16, v, T.GradientDescent(0.0), 32)
    try:
        logZ, grads, raw = tr.gradients(sites, seed=4)
    finally:
        tr.close()
    Q, pi, ll, lr = v.evaluate()
    sub = genome[:, sites, :]
    with _ffi.Context(16, 6, 32) as ctx:
        ctx.set_leaves(sub)
        ctx.set_model(Q, pi, ll, lr)
        out = ctx.sweep(4)
    assert out['logZ'] == logZ
    f = G.forward(sub, Q, pi, ll, lr, 16, 4)
    st = f['struct']
    for r in range(1, 5):
        st['anc'][r] = out['ancestors'][r - 1].astype(np.int64)
    ref = G.to_variables(Q, pi, ll, lr, G.sweep_grad(sub, Q, pi, ll, lr, 16, 4, struct=st))
    for mine, theirs in (('a_l', 'd_loglam_l'), ('a_r', 'd_loglam_r'), ('y_station', 'd_y_station'), ('y_q', 'd_y_q')):
        scale = np.max(np.abs(ref[theirs]))
        assert np.max(np.abs(grads[mine] - ref[theirs])) < 1e-9 * scale, mine


@pytest.mark.parametrize("nested,jc,opt", [(False, False, 'Adam'), (False, True, 'GD'), (True, False, 'Adam')])
def test_training_step_in_the_library_equals_the_numpy_step(nested, jc, opt):
    """Trainer.step with the host half in the library (phylo_vi_gradients / phylo_vi_apply: model from the variables, chain rules,
    optimiser) against the NumPy statements of phylo_amd/train.py: the same variables after four steps (exp of the host's libm
    against NumPy's: a last-bit difference in Q would show as 1e-16 in the gradients, not as a different genealogy here)."""
    from phylo_amd import train as T
    genome = load_dataset('primate_data_wang')['genome'][:, :96]
    N = genome.shape[0]
    out = {}
    for native in (True, False):
        v = T.Variables(N, np.log(10.0), jc)
        tr = T.Trainer(genome, 48, v, T.make_optimizer(opt if opt == 'Adam' else 'GradientDescentOptimizer', 0.02), 96, nested=nested, M=2,
                       native=native)
        costs = [tr.step(np.arange(96), seed=30 + i) for i in range(4)]
        out[native] = (costs, {n: np.array(getattr(v, n)) for n in v.names()}, tr.last['raw'])
        tr.close()
    assert out[True][2]['backward_lists'] == out[False][2]['backward_lists']
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-12)
    for n in out[True][1]:
        np.testing.assert_allclose(out[True][1][n], out[False][1][n], rtol=1e-10, atol=1e-13)


def test_full_size_gradient_is_the_directional_derivative():
    """BASELINE size (primate, K = 2048, all 898 sites): central difference of the forward sweep along the
    gradient direction, same seed (the resampling outcomes must not change for the difference to be smooth)."""
    from phylo_amd import train as T
    genome = load_dataset('primate_data')['genome']
    N, S, _ = genome.shape
    K = 2048
    v = T.Variables(N, np.log(10.0), jcmodel=False)
    tr = T.Trainer(genome, K, v, T.GradientDescent(0.0), S)
    try:
        logZ, grads, raw = tr.gradients(np.arange(S), seed=99)
        names = v.names()
        norm = np.sqrt(sum(np.sum(grads[n] ** 2) for n in names))
        assert np.isfinite(norm) and norm > 0
        base = {n: getattr(v, n).copy() for n in names}
        eps = 1e-6
        vals, ancs = [], []
        for sgn in (+1.0, -1.0):
            for n in names:
                setattr(v, n, base[n] + sgn * eps * grads[n] / norm)
            Q, pi, ll, lr = v.evaluate()
            tr.ctx.set_model(Q, pi, ll, lr)
            o = tr.ctx.sweep(99)
            vals.append(o['logZ'])
            ancs.append(o['ancestors'])
        assert np.array_equal(ancs[0], ancs[1]), "a resampling outcome flipped inside the finite-difference interval"
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - norm) < 1e-4 * norm, (fd, norm)
        assert raw['backward_ms'] > 0
    finally:
        tr.close()


def test_vcsmc_train_takes_optimizer_steps(tmp_path):
    """VCSMC.train with the reference's loop: variables move, rates stay positive, artefacts are written, and
    gradient ascent with a sane step improves the minibatch ELBO on average."""
    from phylo_amd.vcsmc import VCSMC, default_args
    import random
    random.seed(1)
    data = load_dataset('primate_data')
    args = default_args(n_particles=256, optimizer='Adam', learning_rate=0.05, batch_size=256, seed=7)
    v = VCSMC(data, 256, args)
    lam0 = v.left_branches_param.copy()
    elbos = v.train(epochs=6, batch_size=256, learning_rate=0.05, save_dir=str(tmp_path))
    assert len(elbos) == 6 and np.all(np.isfinite(elbos))
    assert len(v.minibatch_costs) == 6 * 3                 # 898 sites: 3 full slices + remainder, last one skipped
    assert not np.array_equal(v.left_branches_param, lam0) and np.all(v.left_branches_param > 0)
    assert not np.allclose(v.Qmatrix, 1 / 3 * (1 - np.eye(4)) - np.eye(4))
    assert np.allclose(v.Qmatrix.sum(axis=1), 0.0, atol=1e-12) and np.isclose(v.stationary_probs.sum(), 1.0)
    assert (tmp_path / 'results.p').exists()
    assert 'AdamOptimizer' in (tmp_path / 'run_parameters.txt').read_text()
    assert np.mean(elbos[-2:]) > np.mean(elbos[:2]), elbos
    v.close()


@pytest.mark.parametrize("N,S,K", [(2, 5, 4), (3, 1, 1), (3, 7, 2), (4, 65, 3)])
def test_gradient_edge_shapes(N, S, K):
    """Two taxa (one rank event, no resampling), one particle, one site, a ragged 64-site group."""
    rng = np.random.default_rng(100 + N * 10 + K)
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    _check(genome, Q, pi, ll, lr, K=K, seed=6)


def test_gradient_same_with_eager_nodes_and_after_sweep_node():
    """The reverse pass of a lazy kept graph (free nodes before the lists, from the marks), of an eager one (no marks), and of a
    lazy one whose marks phylo_sweep_node has widened give the same gradient; more than one 256-site tile per adopted node."""
    rng = np.random.default_rng(41)
    N, S, K = 7, 300, 96
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    g, _ = _check(genome, Q, pi, ll, lr, K=K, seed=21)
    g_eager, _ = _check(genome, Q, pi, ll, lr, K=K, seed=21, flags=_ffi.FLAGS_DEFAULT | _ffi.EAGER_NODES)
    with _ffi.Context(K, N, S) as ctx:
        ctx.set_leaves(genome)
        ctx.set_model(Q, pi, ll, lr)
        ctx.sweep(21, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
        ctx.sweep_node(2, 5)                             # writes (and marks) every node the lazy sweep skipped
        g_node = ctx.sweep_backward()
    for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
        scale = max(np.max(np.abs(g[key])), 1e-300)
        assert np.max(np.abs(g_eager[key] - g[key])) / scale < 1e-12, key
        assert np.max(np.abs(g_node[key] - g[key])) / scale < 1e-12, key


@pytest.mark.parametrize("twisted", [False, True])
def test_gradient_same_on_two_streams(monkeypatch, twisted):
    """Large sweeps run pg_nodes_free in the background and the adopted nodes' chain on a second stream (PHYLO_GRAD_TWO_STREAMS
    forces that at any size; the switches are read by phylo_create): same gradient, bit for bit, as on one stream."""
    rng = np.random.default_rng(45)
    N, S, K = 7, 130, 64
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    run = (lambda: _check_twisted(genome, Q, pi, ll, lr, K=K, M=2, seed=8)) if twisted else (lambda: _check(genome, Q, pi, ll, lr, K=K, seed=8))
    monkeypatch.setenv('PHYLO_GRAD_TWO_STREAMS', '1')
    g2 = run()[0]
    monkeypatch.delenv('PHYLO_GRAD_TWO_STREAMS')
    monkeypatch.setenv('PHYLO_GRAD_ONE_STREAM', '1')
    g1 = run()[0]
    for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
        assert np.array_equal(g1[key], g2[key]), key


def _device_lists_match_host(N, S, K, genome, Q, pi, ll, lr, seed):
    """The lists the device kernels build (phylo_revlists_dev.h) against the host builders on the same graph: identical but for the
    two documented differences -- heavy[] global instead of per rank event, a node's flagged parents ascending instead of descending."""
    with _ffi.Context(K, N, S) as ctx:
        ctx.set_leaves(genome)
        ctx.set_model(Q, pi, ll, lr)
        ctx.sweep(seed, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
        d = ctx.debug_device_lists()
        d2 = ctx.debug_device_lists()
    return _compare_device_lists_with_host(N, K, d, d2)


def _compare_device_lists_with_host(N, K, d, d2):
    h = _ffi.debug_reverse_lists(N, K, d['ancestors'], d['child'], early_free=True, rows_form=True)
    R, nn = N - 1, (N - 1) * K
    for key in ('n_adp', 'n_chunks', 'n_slow', 'n_par'):
        assert d[key] == h[key], key
    for key in ('ev_adp0', 'ev_slow0'):
        assert np.array_equal(d[key], h[key]), key
    assert np.array_equal(d['ad_off'][1:], h['ad_off'][1:])
    assert np.array_equal(d['ad_idx'][1:], h['ad_idx'][1:])
    assert np.array_equal(d['adp'][:h['n_adp']], h['adp'][:h['n_adp']])
    assert np.array_equal(d['par_off'], h['par_off'])
    assert np.array_equal(d['slow_flag'], h['slow_flag'])
    assert np.array_equal(d['slow_idx'][:h['n_slow']], h['slow_idx'][:h['n_slow']])
    assert np.array_equal(d['chunk_beg'][:h['n_chunks']], h['chunk_beg'][:h['n_chunks']])
    assert np.array_equal(d['chunk_cnt'][:h['n_chunks']], h['chunk_cnt'][:h['n_chunks']])
    rc0 = np.repeat(h['rank_chunk0'][:R], K)
    assert np.array_equal(d['heavy'], np.where(h['heavy'] >= 0, h['heavy'] + rc0, -1))
    FREE = 1 << 30
    po = h['par_off']
    n_two = 0
    for x in np.nonzero(np.diff(po))[0]:                # a node's list: the free parents ascending, then the flagged ones
        a, b = d['par_idx'][po[x]:po[x + 1]], h['par_idx'][po[x]:po[x + 1]]
        fa, fb = a[(a & FREE) != 0], b[(b & FREE) != 0]
        assert np.array_equal(fa, fb) and np.array_equal(a[:len(fa)], fa), x
        assert np.array_equal(a[len(fa):], b[len(fb):][::-1]), x
        assert np.all(np.diff(a[len(fa):]) > 0), x
        n_two += len(a) - len(fa) >= 2
    for key in ('ad_idx', 'par_idx', 'slow_idx', 'adp', 'heavy', 'chunk_beg'):   # built twice: the same lists
        assert np.array_equal(d[key], d2[key]), key
    return h, n_two


def test_device_built_lists_equal_the_host_builders_degenerate_genealogy():
    """Real sites: a few ancestors take nearly every draw (heavy nodes, chunks)."""
    genome = load_dataset('primate_data')['genome'][:, :96]
    N = genome.shape[0]
    rng = np.random.default_rng(5)
    Q, pi, ll, lr = _model(rng, N)
    h, _ = _device_lists_match_host(N, 96, 1500, genome, Q, pi, ll, lr, seed=4)   # (K not a multiple of the builders' block)
    assert h['n_chunks'] > 0


def test_device_built_lists_equal_the_host_builders_flat_weights():
    """All-gap rows: hundreds of distinct ancestors per rank event, nodes with several flagged parents."""
    rng = np.random.default_rng(43)
    N, S, K = 9, 40, 700
    Q, pi, ll, lr = _model(rng, N)
    h, n_two = _device_lists_match_host(N, S, K, np.ones((N, S, 4)), Q, pi, ll, lr, seed=3)
    assert h['n_adp'] > K and n_two > 0


@pytest.mark.parametrize("N,K,survivors,seed", [
    (2, 1, 1, 0), (2, 5, 3, 1), (3, 4, 2, 2), (4, 3, 3, 3), (6, 64, 3, 4), (6, 64, 64, 5), (7, 130, 5, 6), (8, 257, 12, 7),
    (12, 512, 4, 8), (5, 1023, 2, 9), (5, 1025, 40, 10), (4, 2100, 7, 11), (3, 4097, 300, 12), (3, 8192, 3, 13),
])
def test_device_built_lists_on_the_random_genealogies_of_the_cpu_tests(N, K, survivors, seed):
    """The cases of tests/test_revlists_cpu.py (random genealogies from flat to degenerate, children from any earlier rank event),
    and more particles than they use -- every workgroup shape of pg_dl_adopters (1, 2, 4, 8 adopters per thread), K not a multiple
    of anything -- through the device builders, against the host builders those tests check against a restatement in Python."""
    from test_revlists_cpu import _random_genealogy          # (pytest puts tests/ on sys.path)
    rng = np.random.default_rng(seed)
    anc, child = _random_genealogy(rng, N, K, survivors) if K <= 1100 else _fast_genealogy(rng, N, K, survivors)
    with _ffi.Context(K, N, 4) as ctx:
        d = ctx.debug_device_lists(anc, child)
        d2 = ctx.debug_device_lists(anc, child)
        with pytest.raises(_ffi.PhyloError):             # the genealogy replaced the sweep's: no reverse pass on it
            ctx.sweep_backward()
    _compare_device_lists_with_host(N, K, d, d2)


def _fast_genealogy(rng, N, K, survivors):
    """_random_genealogy of tests/test_revlists_cpu.py without its Python loops over particles (large K)."""
    R = N - 1
    anc = np.zeros((max(R - 1, 0), K), dtype=np.int64)
    for r in range(R - 1):
        pool = rng.choice(K, size=min(survivors, K), replace=False)
        anc[r] = rng.choice(pool, size=K, p=rng.dirichlet(np.full(len(pool), 0.3)))
    child = rng.integers(0, N, size=(R, K, 2)).astype(np.int32)
    for r in range(1, R):
        internal = rng.random((K, 2)) < 0.5
        rp = rng.integers(0, r, size=(K, 2))
        own = rng.random((K, 2)) < 0.8
        kp = np.where(own & (rp < R - 1), anc[np.minimum(rp, max(R - 2, 0)), np.arange(K)[:, None]] if R > 1 else 0, rng.integers(0, K, size=(K, 2)))
        child[r] = np.where(internal, N + rp * K + kp, child[r])
    return anc, child


@pytest.mark.parametrize("N,S,K", [(2, 5, 4), (3, 7, 2), (4, 65, 3)])
def test_device_built_lists_edge_shapes(N, S, K):
    rng = np.random.default_rng(100 + N * 10 + K)
    Q, pi, ll, lr = _model(rng, N)
    _device_lists_match_host(N, S, K, _codes_genome(rng, N, S), Q, pi, ll, lr, seed=6)


def test_gradient_same_with_host_built_lists(monkeypatch):
    """PHYLO_REV_HOST_LISTS keeps the host builders: the same gradient to the last few bits (flagged parents are added in another
    fixed order), on a degenerate and on a flat genealogy."""
    rng = np.random.default_rng(46)
    N, S, K = 8, 130, 300
    Q, pi, ll, lr = _model(rng, N)
    for genome in (_codes_genome(rng, N, S), np.ones((N, S, 4))):
        g_dev = _check(genome, Q, pi, ll, lr, K=K, seed=8)[0]
        monkeypatch.setenv('PHYLO_REV_HOST_LISTS', '1')
        g_host = _check(genome, Q, pi, ll, lr, K=K, seed=8)[0]
        monkeypatch.delenv('PHYLO_REV_HOST_LISTS')
        for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
            scale = max(np.max(np.abs(g_host[key])), 1e-300)
            assert np.max(np.abs(g_dev[key] - g_host[key])) / scale < 1e-12, key


def test_adopted_nodes_in_one_launch_equal_a_launch_per_rank_event(monkeypatch):
    """The adopted nodes' adjoints in ONE launch (workgroups wait for the tiles of their flagged parents: pg_nodes_rows_all) do the
    arithmetic of the launch per rank event (PHYLO_GRAD_ROWS_CHAIN) in the same order: the same bits, run after run, on a
    degenerate genealogy (few adopted nodes, long lists) and on a flat one (hundreds of adopted nodes per rank event, chains of
    flagged parents through every rank event), rows of three tiles.  The chunk sums in quad form (PHYLO_GRAD_QUAD_CHUNKS) differ
    in the order inside the 4-term products only."""
    rng = np.random.default_rng(47)
    N, S, K = 8, 520, 256
    Q, pi, ll, lr = _model(rng, N)
    for genome in (_codes_genome(rng, N, S), np.ones((N, S, 4))):
        runs = [_check(genome, Q, pi, ll, lr, K=K, seed=11)[0] for _ in range(3)]
        monkeypatch.setenv('PHYLO_GRAD_ROWS_CHAIN', '1')
        chain = _check(genome, Q, pi, ll, lr, K=K, seed=11)[0]
        monkeypatch.delenv('PHYLO_GRAD_ROWS_CHAIN')
        monkeypatch.setenv('PHYLO_GRAD_QUAD_CHUNKS', '1')
        quad = _check(genome, Q, pi, ll, lr, K=K, seed=11)[0]
        monkeypatch.delenv('PHYLO_GRAD_QUAD_CHUNKS')
        for key in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'):
            for other in runs[1:] + [chain]:
                assert np.array_equal(np.asarray(runs[0][key]).view(np.uint64), np.asarray(other[key]).view(np.uint64)), key
            scale = max(np.max(np.abs(quad[key])), 1e-300)
            assert np.max(np.abs(runs[0][key] - quad[key])) / scale < 1e-12, key


def test_gradient_flat_weights_many_adopted_nodes():
    """All-gap rows: every weight equal, so hundreds of distinct ancestors survive each resampling -- many adopted nodes with few
    parents each (the opposite of the degenerate genealogies of real data)."""
    rng = np.random.default_rng(43)
    N, S, K = 8, 70, 256
    genome = np.ones((N, S, 4))
    Q, pi, ll, lr = _model(rng, N)
    _check(genome, Q, pi, ll, lr, K=K, seed=3)


def test_gradient_more_than_4096_sites():
    """S > 4096: the reverse pass takes the tile form (pg_nodes, one workgroup per 256 sites of a node, eager nodes) instead of the
    row form; a heavy node (more than eight parents) so that the parent chunks run too."""
    rng = np.random.default_rng(47)
    N, S, K = 5, 4100, 24
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    _check(genome, Q, pi, ll, lr, K=K, seed=9)


def test_twisted_gradient_more_than_4096_sites():
    rng = np.random.default_rng(48)
    N, S, K = 4, 4100, 6
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    _check_twisted(genome, Q, pi, ll, lr, K=K, M=2, seed=4)


def test_gradient_many_taxa():
    """27 taxa (DS1 sites): root tables longer than one slot group of pg_coeff, deep adoption chains."""
    genome = load_dataset('hohna_data_1')['genome'][:, :130]
    rng = np.random.default_rng(31)
    Q, pi, ll, lr = _model(rng, genome.shape[0], spread=0.2, lam=2.3)
    _check(genome, Q, pi, ll, lr, K=40, seed=12)


@pytest.mark.parametrize("N,S,K,M", [(2, 5, 4, 2), (3, 1, 1, 1), (3, 7, 2, 3), (4, 65, 3, 1)])
def test_twisted_gradient_edge_shapes(N, S, K, M):
    rng = np.random.default_rng(200 + N * 10 + K)
    genome = _codes_genome(rng, N, S)
    Q, pi, ll, lr = _model(rng, N)
    _check_twisted(genome, Q, pi, ll, lr, K=K, M=M, seed=6)


def test_twisted_gradient_more_than_256_rows_per_particle():
    """9 taxa, M = 8: J = 288 sub-samples at the first rank event (pg_twist_finish in slices of 256 rows), few particles (the
    chunks of pg_twist_xchunks are single entries with sliced partner slots)."""
    genome = load_dataset('primate_data_wang')['genome'][:, :40]
    rng = np.random.default_rng(35)
    Q, pi, ll, lr = _model(rng, genome.shape[0], spread=0.2, lam=2.3)
    _check_twisted(genome, Q, pi, ll, lr, K=6, M=8, seed=3)


def test_twisted_gradient_many_taxa():
    genome = load_dataset('hohna_data_1')['genome'][:14, :70]
    rng = np.random.default_rng(33)
    Q, pi, ll, lr = _model(rng, genome.shape[0], spread=0.2, lam=2.3)
    _check_twisted(genome, Q, pi, ll, lr, K=24, M=1, seed=12)


def test_twisted_full_size_gradient_is_the_directional_derivative():
    """primate, K = 2048, all 898 sites, twisted proposal (M = 1): central difference of the forward sweep along the
    gradient, same seed; every discrete outcome (ancestors, chosen pairs) must be the same at both ends."""
    from phylo_amd import train as T
    genome = load_dataset('primate_data')['genome']
    N, S, _ = genome.shape
    K = 2048
    v = T.Variables(N, np.log(10.0), jcmodel=False)
    tr = T.Trainer(genome, K, v, T.GradientDescent(0.0), S, nested=True, M=1)
    try:
        logZ, grads, raw = tr.gradients(np.arange(S), seed=99)
        names = v.names()
        norm = np.sqrt(sum(np.sum(grads[n] ** 2) for n in names))
        assert np.isfinite(norm) and norm > 0
        base = {n: getattr(v, n).copy() for n in names}
        eps = 1e-7
        vals, disc = [], []
        for sgn in (+1.0, -1.0):
            for n in names:
                setattr(v, n, base[n] + sgn * eps * grads[n] / norm)
            Q, pi, ll, lr = v.evaluate()
            tr.ctx.set_model(Q, pi, ll, lr)
            o = tr.ctx.sweep(99, _ffi.FLAGS_DEFAULT | _ffi.TWISTING, 1)
            vals.append(o['logZ'])
            disc.append((o['ancestors'], o['merges']))
        assert np.array_equal(disc[0][0], disc[1][0]) and np.array_equal(disc[0][1], disc[1][1]), \
            "a discrete outcome flipped inside the finite-difference interval"
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - norm) < 1e-3 * norm, (fd, norm)
        print('twisted K=2048 forward %.3f ms, backward %.3f ms' % (raw['forward_ms'], raw['backward_ms']))
    finally:
        tr.close()


def test_vcsmc_train_nested_takes_optimizer_steps(tmp_path):
    """VCSMC.train with args.nested (vncsmc.py:568-640): the twisted proposal's reverse pass drives the optimiser."""
    from phylo_amd.vcsmc import VCSMC, default_args
    import random
    random.seed(2)
    data = load_dataset('primate_data_wang')
    args = default_args(n_particles=64, optimizer='Adam', learning_rate=0.05, batch_size=256, seed=7, nested=True, M=2)
    v = VCSMC(data, 64, args)
    lam0 = v.left_branches_param.copy()
    elbos = v.train(epochs=4, batch_size=256, learning_rate=0.05, save_dir=str(tmp_path))
    assert len(elbos) == 4 and np.all(np.isfinite(elbos))
    assert len(v.minibatch_costs) > 0 and np.all(np.isfinite(v.minibatch_costs))
    assert not np.array_equal(v.left_branches_param, lam0) and np.all(v.left_branches_param > 0)
    assert not np.allclose(v.Qmatrix, 1 / 3 * (1 - np.eye(4)) - np.eye(4))
    v.close()


def test_twisted_gradient_many_sub_samples():
    """M = 40 on 6 taxa: J = 600 sub-samples per particle at the first rank event (three slices of pg_twist_finish)."""
    rng = np.random.default_rng(41)
    genome = _codes_genome(rng, 6, 30)
    Q, pi, ll, lr = _model(rng, 6)
    _check_twisted(genome, Q, pi, ll, lr, K=5, M=40, seed=9)


def test_twisted_gradient_more_sub_samples_than_lds_holds():
    """20 taxa, M = 44: J = 8360 > 8192 sub-samples at the first rank event (pg_twist_tau works in the tau rows, pk_twist_choose in
    its global weight buffer)."""
    genome = load_dataset('hohna_data_1')['genome'][:20, :8]
    rng = np.random.default_rng(43)
    Q, pi, ll, lr = _model(rng, genome.shape[0], spread=0.2, lam=2.3)
    _check_twisted(genome, Q, pi, ll, lr, K=2, M=44, seed=4)
