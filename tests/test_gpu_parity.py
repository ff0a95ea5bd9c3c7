"""GPU parity: the HIP path (through the C ABI, phylo_amd/_ffi.py) against the CPU oracle on the same
seeded inputs.  Bit-exact for everything the arithmetic contract covers (resampling indices, merges,
partials, log-weights, log Z-hat); golden fixtures from the reference's csmc.py at fp tolerance."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cpu_ref as O
from phylo_amd import _ffi
from phylo_amd.datasets import load_dataset, synthetic_alignment

pytestmark = pytest.mark.gpu

PI = np.full((1, 4), 0.25)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    neq = bits(a) != bits(b)
    # NaN payloads aside, every bit must match
    both_nan = np.isnan(a) & np.isnan(b)
    bad = neq & ~both_nan
    assert not bad.any(), "%s: %d of %d values differ; first: gpu=%r cpu=%r" % (
        what, bad.sum(), bad.size, a[bad][:1], b[bad][:1])


def make_ctx(genome, K, Q, lam=10.0, jc=False, pi=PI):
    N, S, _ = genome.shape
    ctx = _ffi.Context(K, N, S)
    ctx.set_leaves(genome)
    ctx.set_model(Q, pi, np.full(N - 1, lam), np.full(N - 1, lam), jc69_closed_form=jc)
    return ctx


@pytest.fixture(scope="module")
def primate():
    return load_dataset('primate_data')['genome']


@pytest.fixture(scope="module")
def small():
    return load_dataset('primate_data_wang')['genome']


def test_device_arithmetic_contract(small):
    """exp / log / division / fma on the GPU are bit-identical to the CPU statement of the contract."""
    ctx = make_ctx(small[:3, :16], 2, O.jc_Q())
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-745, 709, 100000), [0.0, -0.0, 1e-300, -1e-300, 709.7, -745.1, np.inf, -np.inf]])
    assert_bit_equal(ctx.math_probe(0, x), CO.math_probe(0, x), "exp")
    x = np.concatenate([np.exp(rng.uniform(-744, 709, 100000)), rng.uniform(0.5, 2, 50000),
                        [0.0, 1.0, 5e-324, 2.2e-308, np.inf, -1.0]])
    assert_bit_equal(ctx.math_probe(1, x), CO.math_probe(1, x), "log")
    # the branch-free exp of the scan (pm_exp_nonpos) equals pm_exp on its whole domain x <= 0
    x = np.concatenate([-rng.uniform(0, 760, 600000), -np.exp(rng.uniform(-60, 7, 400000)),
                        [0.0, -0.0, -5e-324, -2.2e-308, -3.7252902984619141e-09, -3.72529029846191e-09, -3.7252902984619145e-09,
                         -0.34657359027997264, -0.3465735902799726, -0.3465735902799727, -708.3964185322641, -709.0, -744.44, -745.13,
                         -745.1332191019411, -745.1332191019412, -745.2, -800.0, -1e300, -np.inf]])
    assert_bit_equal(ctx.math_probe(4, x), CO.math_probe(0, x), "exp for non-positive arguments")
    a, b = rng.normal(size=200000) * 10.0 ** rng.integers(-200, 200, 200000), rng.normal(size=200000)
    assert_bit_equal(ctx.math_probe(2, a, b), CO.math_probe(2, a, b), "div")
    assert_bit_equal(ctx.math_probe(3, a, b), CO.math_probe(3, a, b), "fma")
    ctx.close()


def test_expm_batched_bit_exact_and_vs_scipy(golden_dir, small):
    ex = np.load(os.path.join(golden_dir, "expm_tables.npz"))
    t = np.concatenate([ex['t'], np.random.default_rng(1).exponential(0.1, 500)])
    for q in ('csmc', 'jc', 'gtr_init', 'rand0', 'rand1', 'rand2'):
        ctx = make_ctx(small[:3, :16], 2, ex['Q/' + q])
        P = ctx.expm_batched(t)
        assert_bit_equal(P, CO.expm_batched(ex['Q/' + q], t), "expm " + q)
        np.testing.assert_allclose(P[:len(ex['t'])], ex['P/' + q], rtol=0, atol=1e-14)   # scipy.linalg.expm
        ctx.close()
    ctx = make_ctx(small[:3, :16], 2, ex['Q/jc'], jc=True)
    P = ctx.expm_batched(t)
    assert_bit_equal(P, CO.expm_batched(ex['Q/jc'], t, jc=True), "jc closed form")
    np.testing.assert_allclose(P[:len(ex['t'])], ex['P/jc'], rtol=0, atol=3e-15)
    assert ctx.expm_batched(np.zeros(0)).shape == (0, 4, 4)                                # empty input
    ctx.close()


def test_cond_likelihood_K_vs_oracle_and_reference_formula(primate):
    rng = np.random.default_rng(2)
    Q = O.get_Q(rng.normal(size=(4, 4)))
    for K, S in [(1, 1), (7, 130), (33, 898), (5, 257)]:       # ragged: S not a multiple of 256, S = 1
        g = primate[:, :S]
        li, ri = rng.integers(0, 12, K), rng.integers(0, 12, K)
        l = g[li] * rng.uniform(0.1, 1.0, (K, S, 1))           # not just one-hot rows
        r = g[ri]
        tl, tr = rng.exponential(0.1, K), rng.exponential(0.1, K)
        ctx = make_ctx(primate[:, :S], 4, Q)
        out = ctx.cond_likelihood_K(l, r, tl, tr)
        assert_bit_equal(out, CO.cond_likelihood_K(Q, l, r, tl, tr), "cond_likelihood_K")
        np.testing.assert_allclose(out, O.broadcast_conditional_likelihood_K(Q, l, r, tl, tr), rtol=1e-12)
        ctx.close()


def test_forest_loglik_vs_oracle(primate):
    rng = np.random.default_rng(3)
    K, X, S = 9, 5, 898
    core = primate[rng.integers(0, 12, (K, X))] * rng.uniform(0.2, 1.0, (K, X, S, 1))
    rec = rng.integers(1, 6, (K, X)).astype(np.int32)
    ctx = make_ctx(primate, 4, O.jc_Q())
    out = ctx.forest_loglik(core, rec)
    assert_bit_equal(out, CO.forest_loglik(PI, core, rec), "forest_loglik")
    np.testing.assert_allclose(out, O.compute_forest_posterior(PI, core, rec), rtol=1e-12)
    # leaves only: S log(1/4) per non-gap leaf, gaps contribute log 1 = 0
    leaf = ctx.forest_loglik(primate[None, :, :, :], np.ones((1, 12), dtype=np.int32))
    n_nongap = int((primate.sum(axis=2) == 1).sum())
    assert leaf[0] == pytest.approx(n_nongap * np.log(0.25), rel=1e-13)
    ctx.close()


def test_tree_loglik_golden_from_reference(golden_dir):
    nodes = np.load(os.path.join(golden_dir, "csmc_nodes.npz"))
    prior = np.ones(4) / 4
    for tag in nodes['cases']:
        dname, shape, qname = str(tag).split('/')
        g = nodes['genome/' + dname]
        ctx = make_ctx(g, 2, nodes['Q/' + qname])
        la, ra = nodes[tag + '/left'], nodes[tag + '/right']
        ll, data = ctx.tree_loglik(la, ra, nodes[tag + '/bl'], nodes[tag + '/br'], int(nodes[tag + '/root']), g, prior)
        np.testing.assert_allclose(data, nodes[tag + '/root_data'], rtol=1e-12, atol=0)
        assert ll == pytest.approx(float(nodes[tag + '/loglik']), rel=1e-12)
        ll_c, data_c = CO.tree_loglik(nodes['Q/' + qname], prior, la, ra, nodes[tag + '/bl'], nodes[tag + '/br'],
                                      int(nodes[tag + '/root']), g)
        assert_bit_equal(data, data_c, "tree root data " + str(tag))
        assert_bit_equal(ll, ll_c, "tree loglik " + str(tag))
        ctx.close()


def test_resample_bit_exact_and_edge_cases(small):
    ctx = make_ctx(small[:3, :16], 2, O.jc_Q())
    rng = np.random.default_rng(4)
    # up to 4096: one workgroup, cdf in LDS; beyond: several workgroups per group (pp_scan_multi_*: tiles of 2048 weights)
    for K in (1, 2, 255, 256, 1000, 2048, 4096, 4097, 5000, 6144, 12000, 16384, 20000, 100001):
        for scale in (0.5, 30.0, 400.0):
            lw = rng.normal(scale=scale, size=K) - 6000.0
            idx = ctx.resample(lw, seed=11, step=3)
            np.testing.assert_array_equal(idx, CO.resample(lw, 11, 3))
            np.testing.assert_array_equal(idx, O.resample_indices(lw, 11, 3))
            assert idx.min() >= 0 and idx.max() < K
    lw = np.full(64, -np.inf); lw[17] = -5.0                       # one survivor
    assert (ctx.resample(lw, 1, 1) == 17).all()
    lw = rng.normal(size=300); lw[::7] = np.nan                    # NaN weights never selected
    idx = ctx.resample(lw, 5, 2)
    np.testing.assert_array_equal(idx, CO.resample(lw, 5, 2))
    assert not np.isin(idx, np.arange(0, 300, 7)).any()
    lw = np.full(50, -np.inf)                                      # degenerate: uniform
    np.testing.assert_array_equal(ctx.resample(lw, 5, 2), CO.resample(lw, 5, 2))
    # the same edge cases across several workgroups: NaNs, one survivor in the last (ragged) tile, nothing finite
    lw = rng.normal(scale=40.0, size=9000); lw[::11] = np.nan
    np.testing.assert_array_equal(ctx.resample(lw, 5, 2), CO.resample(lw, 5, 2))
    lw = np.full(9000, -np.inf); lw[8999] = 3.0
    assert (ctx.resample(lw, 1, 1) == 8999).all()
    lw = np.full(9000, -np.inf)
    np.testing.assert_array_equal(ctx.resample(lw, 5, 2), CO.resample(lw, 5, 2))
    lw = rng.normal(scale=20, size=(5, 10000)) - 500               # log Z of large rows (the log-normaliser's workgroup alone)
    assert_bit_equal(ctx.log_zsmc(lw), CO.log_zsmc(lw), "log_zsmc, 10000 weights per row")
    # log Z
    lw = rng.normal(scale=20, size=(11, 777)) - 500
    z = ctx.log_zsmc(lw)
    assert_bit_equal(z, CO.log_zsmc(lw), "log_zsmc")
    assert z == pytest.approx(O.compute_log_ZSMC(lw), rel=1e-13)
    assert ctx.log_zsmc(np.zeros((1, 32))) == pytest.approx(0.0, abs=1e-15)    # row 0 contributes 0
    ctx.close()


@pytest.mark.parametrize("dataset,K,jc,seeds", [
    ('primate_data_wang', 16, True, (0, 1, 2)),        # BASELINE config 0 shape: primates_small JC69 K=16
    ('primate_data', 64, False, (0, 5)),               # GTR-init Q, gaps
    ('primate_data', 300, True, (3,)),                 # K not a power of two
])
def test_sweep_bit_exact_vs_oracle(dataset, K, jc, seeds):
    g = load_dataset(dataset)['genome']
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = make_ctx(g, K, Q, jc=jc)
    for seed in seeds:
        out = ctx.sweep(seed)
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed, jc=jc, want_nodes=True)
        np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])     # indices bit-exact
        np.testing.assert_array_equal(out['merges'], ref['merges'])
        for key in ('left_branches', 'right_branches', 'log_likelihood', 'log_weights'):
            assert_bit_equal(out[key], ref[key], key)
        assert_bit_equal(out['logZ'], ref['logZ'], 'logZ')
        for (r, k) in [(0, 0), (N - 2, K - 1), (N // 2, K // 3)]:
            assert_bit_equal(ctx.sweep_node(r, k), ref['nodes'][r, k], "node partial (%d,%d)" % (r, k))
        # against the independent NumPy oracle: same indices, log Z within 1e-9 relative
        ref2 = O.sweep(g, Q, PI, lam, lam, K, seed)
        np.testing.assert_array_equal(out['ancestors'], ref2['ancestors'])
        assert out['logZ'] == pytest.approx(ref2['logZ'], rel=1e-9)
    # determinism: same seed twice -> identical bits
    a, b = ctx.sweep(7), ctx.sweep(7)
    assert_bit_equal(a['log_weights'], b['log_weights'], "determinism")
    ctx.close()


def test_sweep_quirk_flag_and_trained_like_model():
    """General (asymmetric) row-softmax Q, non-uniform pi, per-rank rates; Q1 flag off = log q."""
    g = load_dataset('primate_data')['genome'][:7, 100:500]
    N = 7
    rng = np.random.default_rng(9)
    Q = O.get_Q(rng.normal(size=(4, 4)))
    pi = O.get_stationary_probs(rng.normal(size=4))
    lam_l, lam_r = rng.uniform(3, 20, N - 1), rng.uniform(3, 20, N - 1)
    K = 128
    ctx = _ffi.Context(K, N, g.shape[1])
    ctx.set_leaves(g)
    ctx.set_model(Q, pi, lam_l, lam_r)
    for flags in (1, 0):
        out = ctx.sweep(21, flags=flags)
        ref = CO.sweep(g, Q, pi, lam_l, lam_r, K, 21, flags=flags)
        np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
        assert_bit_equal(out['log_weights'], ref['log_weights'], "log_weights flags=%d" % flags)
        assert_bit_equal(out['logZ'], ref['logZ'], "logZ")
    ctx.close()


def test_sweep_full_size_properties(primate):
    """BASELINE sizes (primate.p, K=2048): size-independent properties of one sweep (checksum of checksums, indices reproduced
    from the returned weights, valid merges).  The bit-for-bit comparison at this size is
    tests/test_gpu_fullsize.py::test_primate_gtr_K2048_ten_seeds."""
    K, N = 2048, 12
    Q = O.get_Q(O.init_y_q())
    ctx = make_ctx(primate, K, Q)
    out = ctx.sweep(0)
    lw = out['log_weights']
    assert np.isfinite(lw).all()
    # log Z recomputed from the returned weights (checksum of checksums)
    assert out['logZ'] == pytest.approx(O.compute_log_ZSMC(lw), rel=1e-13)
    # resampling indices are reproducible from the returned weights alone
    for r in (1, 5, 10):
        np.testing.assert_array_equal(out['ancestors'][r - 1], O.resample_indices(lw[r - 1], 0, r))
    # merges are valid, distinct root-table slots
    for r in range(N - 1):
        m = out['merges'][r]
        assert (m[:, 0] != m[:, 1]).all() and m.min() >= 0 and m.max() < N - r
    # the C oracle at full size on 3 rank-0 quantities is cheap: branch draws are particle-local
    bl, br = O.branch_samples(K, 10.0, 10.0, 0, 0)
    np.testing.assert_allclose(out['left_branches'][0], bl, rtol=1e-14)
    # untrained level (K=512 about -6616 +- 53 for jc=false; larger K is higher)
    assert -6900 < out['logZ'] < -6300
    st = out['stats']
    assert st['units'] == K * 898 * (N - 1) and st['alg_bytes'] == 96 * st['units']
    ctx.close()


def test_synthetic_large_sites():
    """S = 5000 (many column iterations), N = 20."""
    d = synthetic_alignment(20, 5000)
    g = d['genome']
    K = 24
    lam = np.full(19, 10.0)
    ctx = make_ctx(g, K, O.jc_Q(), jc=True)
    out = ctx.sweep(2)
    ref = CO.sweep(g, O.jc_Q(), PI, lam, lam, K, 2, jc=True)
    np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
    assert_bit_equal(out['log_weights'], ref['log_weights'], "log_weights")
    ctx.close()


def test_error_behaviour(small):
    with pytest.raises(_ffi.PhyloError):
        _ffi.Context(4, 1, 10)                      # N < 2
    with pytest.raises(_ffi.PhyloError):
        _ffi.Context(4, 5, 10, A=6)                 # A != 4
    ctx = _ffi.Context(4, 9, 738)
    with pytest.raises(_ffi.PhyloError) as e:
        ctx.sweep(0)                                # before set_leaves / set_model
    assert e.value.code == -6
    ctx.set_leaves(small)
    with pytest.raises(_ffi.PhyloError):
        ctx.set_model(O.jc_Q(), PI, np.zeros(8), np.ones(8))     # non-positive rate
    ctx.close()


@pytest.mark.parametrize("dataset,K,M,jc", [
    ('primate_data_wang', 16, 3, True),
    ('primate_data', 48, 1, False),             # BASELINE config 2 shape: primate.p, GTR-init, twisting
    ('primate_data', 20, 10, False),            # the reference's default M
    ('primate_data', 6, 64, False),             # M = K of the reference's commented DS runs (autorun.sh:9-12): J = 4224 sub-samples
    ('hohna_data_1', 4, 32, False),             # DS1 with M = 32 (autorun.sh:9): J = 11 232 sub-samples, more than LDS holds
])
def test_twisted_sweep_bit_exact_vs_oracle(dataset, K, M, jc):
    """Row T: the twisted / nested proposal of vncsmc.py:295-416."""
    g = load_dataset(dataset)['genome']
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = make_ctx(g, K, Q, jc=jc)
    for seed in (0, 3):
        out = ctx.sweep(seed, flags=_ffi.FLAGS_DEFAULT | _ffi.TWISTING, M=M)
        ref = CO.sweep_twisted(g, Q, PI, lam, lam, K, M, seed, jc=jc, want_nodes=True)
        np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
        np.testing.assert_array_equal(out['merges'], ref['merges'])
        assert (out['merges'][:, :, 0] < out['merges'][:, :, 1]).all()         # pairs r1 < r2
        for key in ('left_branches', 'right_branches', 'log_likelihood', 'log_weights'):
            assert_bit_equal(out[key], ref[key], key)
        assert_bit_equal(out['logZ'], ref['logZ'], 'logZ')
        assert_bit_equal(ctx.sweep_node(N - 2, K - 1), ref['nodes'][N - 2, K - 1], "node partial")
    # the un-twisted sweep still works on the same context afterwards
    a = ctx.sweep(1)
    b = CO.sweep(g, Q, PI, lam, lam, K, 1, jc=jc)
    assert_bit_equal(a['log_weights'], b['log_weights'], "plain sweep after a twisted one")
    with pytest.raises(_ffi.PhyloError):
        ctx.sweep(0, flags=_ffi.FLAGS_DEFAULT | _ffi.TWISTING, M=1025)
    ctx.close()
    # independent NumPy oracle on a small case
    if K <= 20 and M <= 16:
        ref2 = O.sweep_twisted(g, Q, PI, lam, lam, K, M, 3)
        np.testing.assert_array_equal(out['ancestors'], ref2['ancestors'])
        assert out['logZ'] == pytest.approx(ref2['logZ'], rel=1e-9)


def test_many_taxa_bookkeeping_paths():
    """N = 70 > 64: the wave-per-particle bookkeeping loops over slots in strides of 64; leaf rows that are
    neither one-hot nor all-ones disable the leaf-code fast path (generic rows)."""
    d = synthetic_alignment(70, 40)
    g = d['genome'].copy()
    K = 12
    lam = np.full(69, 10.0)
    for generic in (False, True):
        if generic:
            g[3, 5] = [0.5, 0.5, 0.0, 0.0]                   # an ambiguity row: not codable
        ctx = make_ctx(g, K, O.jc_Q(), jc=True)
        out = ctx.sweep(11)
        ref = CO.sweep(g, O.jc_Q(), PI, lam, lam, K, 11, jc=True)
        np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
        np.testing.assert_array_equal(out['merges'], ref['merges'])
        assert_bit_equal(out['log_weights'], ref['log_weights'], "log_weights N=70 generic=%s" % generic)
        assert_bit_equal(out['logZ'], ref['logZ'], "logZ")
        ctx.close()


def test_leaf_code_path_equals_generic_path(primate, monkeypatch):
    """The 1-byte leaf codes are an access-path optimisation only: same bits as reading the rows."""
    Q = O.get_Q(O.init_y_q())
    tw = _ffi.FLAGS_DEFAULT | _ffi.TWISTING
    a = make_ctx(primate, 96, Q)
    ra = a.sweep(5)
    ta = a.sweep(5, flags=tw, M=2)
    a.close()
    monkeypatch.setenv("PHYLO_NO_LEAF_CODES", "1")
    b = make_ctx(primate, 96, Q)
    rb = b.sweep(5)
    tb = b.sweep(5, flags=tw, M=2)
    b.close()
    assert_bit_equal(ra['log_weights'], rb['log_weights'], "coded vs generic leaves")
    np.testing.assert_array_equal(ra['ancestors'], rb['ancestors'])
    # twisting: leaf-leaf potentials are priced by code pair whenever the DATA is coded, whichever access path runs
    assert_bit_equal(ta['log_weights'], tb['log_weights'], "twisted, coded vs generic leaves")
    np.testing.assert_array_equal(ta['merges'], tb['merges'])


def test_hohna_ds1_config4_shape():
    """BASELINE config 3 shape on one GPU: DS1 (27 taxa, 1949 sites, gaps), K = 256 against the oracle."""
    g = load_dataset('hohna_data_1')['genome']
    N = g.shape[0]
    lam = np.full(N - 1, 10.0)
    Q = O.get_Q(O.init_y_q())
    ctx = make_ctx(g, 256, Q)
    out = ctx.sweep(0)
    ref = CO.sweep(g, Q, PI, lam, lam, 256, 0)
    np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
    assert_bit_equal(out['log_weights'], ref['log_weights'], "DS1 log_weights")
    assert_bit_equal(out['logZ'], ref['logZ'], "DS1 logZ")
    ctx.close()


@pytest.mark.parametrize("N,S,K", [(2, 1, 1), (2, 5, 7), (3, 1, 4), (5, 300, 3000), (4, 17, 9000)])
def test_sweep_edge_shapes(N, S, K):
    """Smallest trees (one rank event, no resampling), single site, single particle, K beyond one scan tile
    (2048) and beyond the LDS-staged scan (8192)."""
    g = synthetic_alignment(N, S, seed=N * 1000 + S)['genome']
    lam = np.full(N - 1, 10.0)
    Q = O.get_Q(O.init_y_q())
    ctx = make_ctx(g, K, Q)
    out = ctx.sweep(2)
    ref = CO.sweep(g, Q, PI, lam, lam, K, 2)
    np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
    np.testing.assert_array_equal(out['merges'], ref['merges'])
    assert_bit_equal(out['log_weights'], ref['log_weights'], "log_weights")
    assert_bit_equal(out['logZ'], ref['logZ'], "logZ")
    if N >= 3:
        tw = ctx.sweep(2, flags=_ffi.FLAGS_DEFAULT | _ffi.TWISTING, M=2)
        rt = CO.sweep_twisted(g, Q, PI, lam, lam, K, 2, 2)
        np.testing.assert_array_equal(tw['ancestors'], rt['ancestors'])
        assert_bit_equal(tw['log_weights'], rt['log_weights'], "twisted log_weights")
    ctx.close()


def test_fused_scan_bookkeeping_launch(primate, monkeypatch):
    """Opt-in single-launch scan + bookkeeping (bounded flag hand-off inside the launch): same bits."""
    Q = O.get_Q(O.init_y_q())
    a = make_ctx(primate, 200, Q)
    ra = a.sweep(9)
    a.close()
    monkeypatch.setenv("PHYLO_FUSE_SCAN", "1")            # the switches are read when a context is created
    b = make_ctx(primate, 200, Q)
    for seed in (9, 10, 11):
        rb = b.sweep(seed)
        if seed == 9:
            assert_bit_equal(ra['log_weights'], rb['log_weights'], "fused vs separate launches")
            np.testing.assert_array_equal(ra['ancestors'], rb['ancestors'])
    b.close()


def test_lazy_nodes_equal_eager_nodes(monkeypatch):
    """Lazy nodes (only nodes whose creator survives the next resampling are written) are an access-path
    optimisation: every output, and every node partial fetched afterwards, has the same bits."""
    g = load_dataset('primate_data')['genome']
    N = 12
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    K = 160
    monkeypatch.setenv("PHYLO_LAZY_NODES", "1")
    ctx = make_ctx(g, K, Q)
    for seed in (0, 1):
        out = ctx.sweep(seed)
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed, want_nodes=True)
        np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
        assert_bit_equal(out['log_weights'], ref['log_weights'], "lazy log_weights")
        assert_bit_equal(out['logZ'], ref['logZ'], "lazy logZ")
        for (r, k) in [(0, 0), (3, 17), (N - 2, K - 1), (5, 100)]:        # dead and live nodes alike
            assert_bit_equal(ctx.sweep_node(r, k), ref['nodes'][r, k], "lazy node (%d,%d)" % (r, k))
    ctx.close()
    # large-S configuration where lazy nodes are the default
    monkeypatch.delenv("PHYLO_LAZY_NODES")
    d = synthetic_alignment(6, 9000)
    ctx = make_ctx(d['genome'], 24, Q)
    out = ctx.sweep(4)
    ref = CO.sweep(d['genome'], Q, PI, np.full(5, 10.0), np.full(5, 10.0), 24, 4, want_nodes=True)
    assert_bit_equal(out['log_weights'], ref['log_weights'], "lazy default, S=9000")
    assert_bit_equal(ctx.sweep_node(2, 5), ref['nodes'][2, 5], "node")
    ctx.close()


def test_interleaved_contexts_are_deterministic(primate):
    """Three contexts on three streams (the bench's configuration), sweeps in flight together: every repetition
    of a seed must give the same bits (a race between kernels of different sweeps would show here)."""
    Q = O.get_Q(O.init_y_q())
    ctxs = [make_ctx(primate, 1024, Q) for _ in range(3)]
    ref = {}
    for rep in range(8):
        for i, c in enumerate(ctxs):
            c.sweep_async(100 + (i + rep) % 3)
        for i, c in enumerate(ctxs):
            out = c.sweep_fetch()
            seed = 100 + (i + rep) % 3
            key = (out['logZ'], out['log_weights'].tobytes(), out['ancestors'].tobytes())
            assert ref.setdefault(seed, key) == key, "sweep with seed %d changed between repetitions" % seed
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("jc,G,Kg", [(True, 3, 32), (False, 4, 48)])
def test_batched_independent_sweeps_equal_single_sweeps(primate, jc, G, Kg):
    """phylo_sweep_batch_async: G sweeps in one set of launches; group g is bit for bit the sweep of Kg particles with
    seeds[g] (own draws, own resampling segment, own log Z-hat)."""
    g = primate
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    seeds = [11, 5, 902, 77][:G]
    ctx = make_ctx(g, G * Kg, Q, jc=jc)
    for rep in range(2):                                   # the second batch reuses every buffer
        ctx.sweep_batch_async([s + rep for s in seeds])
        out = ctx.sweep_fetch()
        logz = ctx.sweep_fetch_logz(G)
        for i, s in enumerate(seeds):
            ref = CO.sweep(g, Q, PI, lam, lam, Kg, s + rep, jc=jc)
            sl = slice(i * Kg, (i + 1) * Kg)
            np.testing.assert_array_equal(out['ancestors'][:, sl], ref['ancestors'])
            np.testing.assert_array_equal(out['merges'][:, sl], ref['merges'])
            for key in ('log_weights', 'log_likelihood', 'left_branches', 'right_branches'):
                assert_bit_equal(out[key][:, sl], ref[key], "%s of batched sweep %d" % (key, i))
            assert logz[i] == ref['logZ']
        assert out['logZ'] == logz[0]
    # a plain sweep on the same context afterwards is the K-particle sweep again
    a = ctx.sweep(3)
    b = CO.sweep(g, Q, PI, lam, lam, G * Kg, 3, jc=jc)
    assert_bit_equal(a['log_weights'], b['log_weights'], "plain sweep after a batch")
    with pytest.raises(_ffi.PhyloError):
        ctx.sweep_batch_async([1, 2, 3, 4, 5, 6, 7][:5] if (G * Kg) % 5 else [1] * 7)   # K not divisible by G
    with pytest.raises(_ffi.PhyloError):
        ctx.sweep_batch_async(seeds, flags=_ffi.FLAGS_DEFAULT | _ffi.TWISTING)
    ctx.close()


@pytest.mark.parametrize("N,S,G,Kg", [(3, 5, 4, 1), (4, 70, 2, 3), (9, 300, 5, 7)])
def test_batched_sweeps_edge_shapes(small, N, S, G, Kg):
    """One particle per group, group sizes that are no multiple of anything, few taxa."""
    g = small[:N, :S]
    lam = np.full(N - 1, 10.0)
    Q = O.get_Q(O.init_y_q())
    ctx = make_ctx(g, G * Kg, Q)
    seeds = [3 + 5 * i for i in range(G)]
    ctx.sweep_batch_async(seeds)
    out = ctx.sweep_fetch()
    logz = ctx.sweep_fetch_logz(G)
    for i, s in enumerate(seeds):
        ref = CO.sweep(g, Q, PI, lam, lam, Kg, s)
        sl = slice(i * Kg, (i + 1) * Kg)
        np.testing.assert_array_equal(out['ancestors'][:, sl], ref['ancestors'])
        assert_bit_equal(out['log_weights'][:, sl], ref['log_weights'], "log_weights of group %d" % i)
        assert logz[i] == ref['logZ']
    ctx.close()


def test_randomised_shapes_and_modes_against_the_oracle():
    """60 random (N, S, K, model, flags, form) draws: plain / batched sweeps, lazy / eager nodes, JC69 closed form /
    expm, Q1 quirk on / off, coded / generic leaves -- every one bit-exact against the C oracle."""
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        N = int(rng.integers(2, 15))
        S = int(rng.choice([1, 3, 63, 64, 65, 127, 256, 257, 300, 513, 900]))
        G = int(rng.choice([1, 1, 2, 3, 5]))
        Kg = int(rng.choice([1, 2, 7, 16, 33, 64, 100]))
        jc = bool(rng.integers(0, 2))
        q1 = bool(rng.integers(0, 2))
        eager = bool(rng.integers(0, 2))
        generic = trial % 7 == 0
        if generic:
            g = rng.uniform(0.05, 1.0, size=(N, S, 4))
        else:
            codes = rng.integers(0, 5, size=(N, S))
            g = np.zeros((N, S, 4))
            for a in range(4):
                g[..., a] = (codes == a) | (codes == 4)
        if jc:
            Q = O.jc_Q()
        else:
            y = rng.normal(size=(4, 4)) * 0.4
            np.fill_diagonal(y, 0.0)
            Q = O.get_Q(y)
        p = np.exp(rng.normal(size=4) * 0.3)
        pi = (p / p.sum())[None, :]
        lam_l = np.exp(rng.normal(size=N - 1) * 0.3 + 2.0)
        lam_r = np.exp(rng.normal(size=N - 1) * 0.3 + 2.0)
        flags = (_ffi.QUIRK_Q1_RAW_Q if q1 else 0) | (_ffi.EAGER_NODES if eager else 0)
        oflags = O.QUIRK_Q1_RAW_Q if q1 else 0
        seeds = [int(rng.integers(0, 2 ** 40)) for _ in range(G)]
        what = "trial %d N=%d S=%d G=%d Kg=%d jc=%s q1=%s eager=%s generic=%s" % (trial, N, S, G, Kg, jc, q1, eager, generic)
        with _ffi.Context(G * Kg, N, S) as ctx:
            ctx.set_leaves(g)
            ctx.set_model(Q, pi, lam_l, lam_r, jc69_closed_form=jc)
            if G == 1:
                out = ctx.sweep(seeds[0], flags=flags)
                logz = [out['logZ']]
            else:
                ctx.sweep_batch_async(seeds, flags=flags)
                out = ctx.sweep_fetch()
                logz = list(ctx.sweep_fetch_logz(G))
            node = ctx.sweep_node(N - 2, G * Kg - 1)
        for i, s in enumerate(seeds):
            ref = CO.sweep(g, Q, pi, lam_l, lam_r, Kg, s, flags=oflags, jc=jc, want_nodes=(i == G - 1))
            sl = slice(i * Kg, (i + 1) * Kg)
            np.testing.assert_array_equal(out['ancestors'][:, sl], ref['ancestors'], err_msg=what)
            np.testing.assert_array_equal(out['merges'][:, sl], ref['merges'], err_msg=what)
            assert_bit_equal(out['log_weights'][:, sl], ref['log_weights'], what)
            assert logz[i] == ref['logZ'] or (np.isnan(logz[i]) and np.isnan(ref['logZ'])), what
        assert_bit_equal(node, ref['nodes'][N - 2, Kg - 1], what + " last node")


# ---- contract v5: the site tile of the canonical sum over sites ------------------------------------------------------

@pytest.mark.parametrize("S,T,K,jc", [
    (738, 64, 24, True),         # 12 tiles, the last one 34 sites
    (738, 128, 24, False),       # 6 tiles
    (449, 192, 40, False),       # tiles of three site steps, the last one 65 sites
    (64, 64, 8, True),           # exactly one tile
    (65, 64, 8, True),           # a second tile of one site
    (9000, 0, 16, False),        # the default tile (2048): five tiles
])
def test_site_tiles_every_form_vs_oracle(S, T, K, jc):
    """Rows longer than one tile: the merge leaves one value per (particle, tile), added left to right (pk_tile_epilogue, or in
    the wave of the one-launch sweep and of the look-ahead potentials).  Lazy, eager, one launch and the twisted proposal,
    each bit for bit against the C oracle run with the same tile."""
    g = (load_dataset('primate_data_wang')['genome'][:7, :S] if S <= 738 else synthetic_alignment(9, S)['genome'])
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = make_ctx(g, K, Q, jc=jc)
    assert _ffi.load().phylo_site_tile(S) == CO.site_tile(S)            # one default on both sides
    ctx.set_site_tile(T)
    CO.set_site_tile(T)
    try:
        assert ctx.site_tile() == CO.site_tile(S)
        ref = CO.sweep(g, Q, PI, lam, lam, K, 3, jc=jc, want_nodes=True)
        for flags in (1, 1 | 8, 1 | 32):                                # lazy, PHYLO_EAGER_NODES, PHYLO_ONE_LAUNCH
            out = ctx.sweep(3, flags=flags)
            np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
            for key in ('log_likelihood', 'log_weights'):
                assert_bit_equal(out[key], ref[key], "%s flags=%d" % (key, flags))
            assert_bit_equal(out['logZ'], ref['logZ'], 'logZ flags=%d' % flags)
            assert_bit_equal(ctx.sweep_node(N - 2, K - 1), ref['nodes'][N - 2, K - 1], "node")
        if S <= 738:
            M = 2
            out = ctx.sweep(5, flags=1 | 2, M=M)
            ref = CO.sweep_twisted(g, Q, PI, lam, lam, K, M, 5, jc=jc)
            np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
            np.testing.assert_array_equal(out['merges'], ref['merges'])
            assert_bit_equal(out['log_weights'], ref['log_weights'], "twisted log_weights")
            # op-level rows through the same tiles
            core = np.random.default_rng(1).uniform(1e-3, 1.0, size=(3, 2, S, 4))
            rec = np.array([[1, 2], [3, 1], [2, 2]], dtype=np.int32)
            assert_bit_equal(ctx.forest_loglik(core, rec), CO.forest_loglik(PI, core, rec), "forest_loglik")
    finally:
        CO.set_site_tile(0)
        ctx.close()
