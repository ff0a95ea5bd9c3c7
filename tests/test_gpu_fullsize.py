"""Full-size parity on the BASELINE configurations, and the one-launch form of the sweep.

BASELINE.json configs: primate.p JC69 K=512; primate.p GTR-init K=2048 (plain, 10 seeds: the |delta log Z|
half of the metric) and with twisting; DS1 K=4096; synthetic 128 x 50 000.  Every comparison is bit for bit against the C
oracle (oracle/csrc/oracle.c: the reference's dataflow, vcsmc.py:332-451, under the arithmetic and RNG contract of
DESIGN.md section 3) on the same seeds; where the oracle would take minutes the workload is cut in K and the full-K run is
checked through size-independent properties."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cpu_ref as O
from phylo_amd import _ffi
from phylo_amd.datasets import load_dataset, synthetic_alignment

pytestmark = pytest.mark.gpu
PI = np.full((1, 4), 0.25)
# the two forms of the whole sweep: launches per rank event (scan, bookkeeping, materialise, merge) and
# one launch of resident workgroups for the whole sweep
FORMS = [_ffi.FLAGS_DEFAULT, _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH]
FORM_IDS = ["launches", "one-launch"]


def ctx_for(g, K, Q, jc=False):
    N, S, _ = g.shape
    ctx = _ffi.Context(K, N, S)
    ctx.set_leaves(g)
    ctx.set_model(Q, PI, np.full(N - 1, 10.0), np.full(N - 1, 10.0), jc69_closed_form=jc)
    return ctx


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a, dtype=np.float64).view(np.uint64),
                          np.ascontiguousarray(b, dtype=np.float64).view(np.uint64))


def check(out, ref, what):
    np.testing.assert_array_equal(out['ancestors'], ref['ancestors'], err_msg=what + ": resampling indices")
    np.testing.assert_array_equal(out['merges'], ref['merges'], err_msg=what + ": merges")
    for key in ('log_weights', 'log_likelihood', 'left_branches', 'right_branches'):
        assert same_bits(out[key], ref[key]), "%s: %s differs" % (what, key)
    assert out['logZ'] == ref['logZ'], (what, out['logZ'], ref['logZ'])


@pytest.mark.parametrize("flags", FORMS, ids=FORM_IDS)
def test_primate_gtr_K2048_ten_seeds(flags):
    """BASELINE config 2 (headline): primate.p, jcmodel=false initial Q, K = 2048, seeds 0..9.  max |delta log Z| = 0 and
    identical ancestor indices at every rank event (SURVEY 8d asks <= 1e-6 |log Z|)."""
    g = load_dataset('primate_data')['genome']
    Q = O.get_Q(O.init_y_q())
    lam = np.full(11, 10.0)
    ctx = ctx_for(g, 2048, Q)
    worst = 0.0
    for seed in range(10):
        out = ctx.sweep(seed, flags=flags)
        ref = CO.sweep(g, Q, PI, lam, lam, 2048, seed)
        check(out, ref, "primate GTR K=2048 seed %d" % seed)
        worst = max(worst, abs(out['logZ'] - ref['logZ']))
    assert worst == 0.0
    ctx.close()


@pytest.mark.parametrize("flags", FORMS, ids=FORM_IDS)
def test_primate_jc69_K512(flags):
    """BASELINE config 1: primate.p, JC69 closed form, K = 512, seeds 0..9."""
    g = load_dataset('primate_data')['genome']
    lam = np.full(11, 10.0)
    ctx = ctx_for(g, 512, O.jc_Q(), jc=True)
    for seed in range(10):
        check(ctx.sweep(seed, flags=flags), CO.sweep(g, O.jc_Q(), PI, lam, lam, 512, seed, jc=True), "primate JC69 K=512 seed %d" % seed)
    ctx.close()


def test_primate_gtr_twisting_K2048():
    """BASELINE config 2 with the twisted proposal (vncsmc.py:295-416), M = 1, K = 2048, two seeds; M = 10 (the reference's
    default) at K = 256."""
    g = load_dataset('primate_data')['genome']
    Q = O.get_Q(O.init_y_q())
    lam = np.full(11, 10.0)
    tw = _ffi.FLAGS_DEFAULT | _ffi.TWISTING
    ctx = ctx_for(g, 2048, Q)
    for seed in (0, 1):
        check(ctx.sweep(seed, flags=tw, M=1), CO.sweep_twisted(g, Q, PI, lam, lam, 2048, 1, seed), "twisting M=1 K=2048 seed %d" % seed)
    ctx.close()
    ctx = ctx_for(g, 256, Q)
    check(ctx.sweep(3, flags=tw, M=10), CO.sweep_twisted(g, Q, PI, lam, lam, 256, 10, 3), "twisting M=10 K=256")
    ctx.close()


@pytest.mark.parametrize("flags", FORMS, ids=FORM_IDS)
def test_ds1_K4096_one_gpu(flags):
    """BASELINE config 3's workload on one GPU: DS1 (27 taxa, 1949 sites, 10 746 gap cells), K = 4096."""
    g = load_dataset('hohna_data_1')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, 4096, Q)
    for seed in (0, 1):
        check(ctx.sweep(seed, flags=flags), CO.sweep(g, Q, PI, lam, lam, 4096, seed), "DS1 K=4096 seed %d" % seed)
    ctx.close()


def test_synthetic_128x50k():
    """BASELINE config 4's workload: synthetic 128 taxa x 50 000 sites (numpy default_rng(20260005), no gaps).  K = 16 bit
    for bit against the oracle (the oracle's K-replicated core is 128 x 50 000 x 32 B per particle); K = 128 through
    size-independent properties: log Z recomputed from the returned weights, resampling indices reproduced from the
    returned weights, valid merges, the first rank event's node against the explicit-tree op."""
    d = synthetic_alignment(128, 50000)
    g = d['genome']
    N, S, _ = g.shape
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, 16, Q)
    check(ctx.sweep(5), CO.sweep(g, Q, PI, lam, lam, 16, 5), "synthetic 128x50k K=16")
    ctx.close()
    K = 128
    ctx = ctx_for(g, K, Q)
    out = ctx.sweep(1)
    lw = out['log_weights']
    assert np.isfinite(lw).all()
    assert out['logZ'] == pytest.approx(O.compute_log_ZSMC(lw), rel=1e-13)
    for r in (1, 2, 64, 126):
        np.testing.assert_array_equal(out['ancestors'][r - 1], O.resample_indices(lw[r - 1], 1, r))
    for r in range(N - 1):
        m = out['merges'][r]
        assert (m[:, 0] != m[:, 1]).all() and m.min() >= 0 and m.max() < N - r
    # the node particle 0 creates at rank event 0 is a cherry of two leaves: compare with the explicit-tree op
    il, ir = out['merges'][0, 0]
    bl, br = out['left_branches'][0, 0], out['right_branches'][0, 0]
    node = ctx.sweep_node(0, 0)
    _, root = ctx.tree_loglik(np.array([-1, -1, 0], dtype=np.int32), np.array([-1, -1, 1], dtype=np.int32),
                              np.array([0, 0, bl]), np.array([0, 0, br]), 2, g[[il, ir]], PI)
    assert same_bits(node, root)
    ctx.close()


# ---- the one-launch form (phylo_persist.h) on shapes that exercise its own paths -------------------------------------

@pytest.mark.parametrize("dataset,K,jc", [
    ('primate_data_wang', 16, True),        # one particle per workgroup
    ('primate_data', 257, False),           # K prime: ONE workgroup owns all particles, 17 chunks of 16
    ('primate_data', 5000, False),          # 250 workgroups x 20 particles (two chunks), three scan tiles
    ('primate_data', 8192, False),          # the largest group whose cdf lives in LDS
    ('hohna_data_1', 300, False),           # N = 27: history rows and root slots fill 27 lanes
])
def test_one_launch_sweep_shapes(dataset, K, jc):
    g = load_dataset(dataset)['genome']
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, K, Q, jc=jc)
    one = _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH
    for seed in (0, 4):
        four = ctx.sweep(seed)                                  # launches per rank event
        assert four['stats']['n_launches'] > 3 * (N - 1)
        out = ctx.sweep(seed, flags=one)
        assert out['stats']['n_launches'] == 1, "the one-launch form did not run"
        check(four, out, "launches vs one launch, %s K=%d seed %d" % (dataset, K, seed))
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed, jc=jc, want_nodes=(K <= 300))
        check(out, ref, "%s K=%d seed %d" % (dataset, K, seed))
        if K <= 300:                                            # dead and adopted nodes alike, written on demand afterwards
            for (r, k) in [(0, 0), (N - 2, K - 1), (N // 2, K // 3)]:
                assert same_bits(ctx.sweep_node(r, k), ref['nodes'][r, k]), "node (%d,%d)" % (r, k)
    # back to launches on the same context, and again one launch: the monotone arrival counters carry over
    check(ctx.sweep(9), CO.sweep(g, Q, PI, lam, lam, K, 9, jc=jc), "launch path after one-launch sweeps")
    check(ctx.sweep(9, flags=one), CO.sweep(g, Q, PI, lam, lam, K, 9, jc=jc), "one-launch again")
    ctx.close()


def test_whole_sweep_forms_quirk_flag_generic_rows_and_special_values():
    """Asymmetric Q, non-uniform pi, per-rank rates, log-q form; a leaf row that is neither one-hot nor all-ones (no leaf
    codes); an all-zero leaf row (site likelihood 0 -> log = -inf: the merge's out-of-line branch for factors that are not
    positive normal numbers)."""
    g = load_dataset('primate_data')['genome'][:7, 100:500].copy()
    N = 7
    rng = np.random.default_rng(9)
    Q = O.get_Q(rng.normal(size=(4, 4)))
    pi = O.get_stationary_probs(rng.normal(size=4))
    lam_l, lam_r = rng.uniform(3, 20, N - 1), rng.uniform(3, 20, N - 1)
    K = 128
    for variant in ('coded', 'generic', 'zero-row'):
        if variant == 'generic':
            g[3, 5] = [0.5, 0.5, 0.0, 0.0]
        if variant == 'zero-row':
            g[2, 7] = [0.0, 0.0, 0.0, 0.0]
        ctx = _ffi.Context(K, N, g.shape[1])
        ctx.set_leaves(g)
        ctx.set_model(Q, pi, lam_l, lam_r)
        for flags in (1, 0, 1 | _ffi.ONE_LAUNCH, 0 | _ffi.ONE_LAUNCH):
            out = ctx.sweep(21, flags=flags)
            if flags & _ffi.ONE_LAUNCH:
                assert out['stats']['n_launches'] == 1
            ref = CO.sweep(g, Q, pi, lam_l, lam_r, K, 21, flags=flags & 1)
            np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
            lw, rw = out['log_weights'], ref['log_weights']
            both_nan = np.isnan(lw) & np.isnan(rw)
            assert (both_nan | (lw.view(np.uint64) == rw.view(np.uint64))).all(), variant
            assert (out['logZ'] == ref['logZ']) or (np.isnan(out['logZ']) and np.isnan(ref['logZ'])), variant
        if variant == 'zero-row':
            assert not np.isfinite(out['log_weights']).all()      # the special values really occurred
        ctx.close()


@pytest.mark.parametrize("G,Kg", [(3, 32), (8, 256), (5, 7), (10, 2048), (13, 1000)])
def test_batched_sweeps_in_every_form(G, Kg):
    """G independent sweeps per set of launches (one launch: every group has its own arrival counter; launches per rank
    event: one scan workgroup per group): each group is bit for bit the Kg-particle sweep of its seed.  (The two largest
    shapes take the prologue whose draws are sorted by Pade order, the last one with a partly filled last workgroup.)"""
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, G * Kg, Q)
    seeds = [100 + 7 * i for i in range(G)]
    for rep, fl in enumerate((_ffi.ONE_LAUNCH, 0, _ffi.ONE_LAUNCH)):
        ctx.sweep_batch_async(seeds, flags=_ffi.FLAGS_DEFAULT | fl)
        out = ctx.sweep_fetch()
        if fl == _ffi.ONE_LAUNCH:
            assert out['stats']['n_launches'] == 1
        logz = ctx.sweep_fetch_logz(G)
        refs = [CO.sweep(g, Q, PI, lam, lam, Kg, s) for s in seeds]
        for key in ('log_weights', 'log_likelihood'):
            assert same_bits(out[key], np.concatenate([r[key] for r in refs], axis=1)), key
        np.testing.assert_array_equal(out['ancestors'], np.concatenate([r['ancestors'] for r in refs], axis=1))
        np.testing.assert_array_equal(out['merges'], np.concatenate([r['merges'] for r in refs], axis=1))
        assert list(logz) == [r['logZ'] for r in refs]
    ctx.close()


def test_large_launch_forms_equal_the_small_launch_forms(monkeypatch):
    """Launch sets of many particles take the 8-lane bookkeeping and the prologue sorted by Pade order; the switches bring back
    the forms small launches use (16 lanes, index order).  Same bits either way."""
    g = load_dataset('primate_data')['genome']
    Q = O.get_Q(O.init_y_q())
    G, Kg = 13, 1000
    seeds = [5 + 3 * i for i in range(G)]
    outs = []
    for switches in ((), ("PHYLO_BOOK_LP16", "PHYLO_NO_SORTED_DRAWS")):
        for sw in switches:
            monkeypatch.setenv(sw, "1")                    # (read when the context is created)
        ctx = ctx_for(g, G * Kg, Q)
        ctx.sweep_batch_async(seeds, flags=_ffi.FLAGS_DEFAULT)
        out = ctx.sweep_fetch()
        outs.append((out, list(ctx.sweep_fetch_logz(G))))
        ctx.close()
    (a, za), (b, zb) = outs
    for key in ('log_weights', 'log_likelihood', 'left_branches', 'right_branches'):
        if key in a:
            assert same_bits(a[key], b[key]), key
    np.testing.assert_array_equal(a['ancestors'], b['ancestors'])
    np.testing.assert_array_equal(a['merges'], b['merges'])
    assert za == zb


def test_one_launch_batches_of_different_G_on_one_context():
    """The one-launch sweep's arrival counters are per group and monotone: a batch with another number of groups that plans
    the same workgroups per group must not find stale counters in the groups the previous batch did not use (K = 6144: G = 3
    then G = 4 then G = 3 again; every batch bit for bit its oracle sweeps, the batched log Z fetch included)."""
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    K = 6144
    ctx = ctx_for(g, K, Q)
    for G in (3, 4, 3, 6):
        Kg = K // G
        seeds = [11 * G + i for i in range(G)]
        ctx.sweep_batch_async(seeds, flags=_ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH)
        logz = ctx.sweep_fetch_logz(G)
        out = ctx.sweep_fetch()
        assert out['stats']['n_launches'] == 1
        refs = [CO.sweep(g, Q, PI, lam, lam, Kg, s) for s in seeds]
        assert same_bits(out['log_weights'], np.concatenate([r['log_weights'] for r in refs], axis=1)), "G=%d" % G
        np.testing.assert_array_equal(out['ancestors'], np.concatenate([r['ancestors'] for r in refs], axis=1))
        assert list(logz) == [r['logZ'] for r in refs]
    ctx.close()


def test_one_launch_is_deterministic_with_contexts_in_flight():
    """Three contexts in flight, each a one-launch sweep of 256 resident workgroups: every repetition of a seed gives the same
    bits (a stale read across workgroups or a lost arrival would show here)."""
    g = load_dataset('primate_data')['genome']
    Q = O.get_Q(O.init_y_q())
    one = _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH
    ctxs = [ctx_for(g, 2048, Q) for _ in range(3)]
    ref = {}
    for rep in range(20):
        for i, c in enumerate(ctxs):
            c.sweep_async(100 + (i + rep) % 3, flags=one)
        for i, c in enumerate(ctxs):
            out = c.sweep_fetch()
            seed = 100 + (i + rep) % 3
            key = (out['logZ'], out['log_weights'].tobytes(), out['ancestors'].tobytes())
            assert ref.setdefault(seed, key) == key, "seed %d changed between repetitions" % seed
    lam = np.full(11, 10.0)
    assert ref[100][0] == CO.sweep(g, Q, PI, lam, lam, 2048, 100)['logZ']
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("flags", FORMS, ids=FORM_IDS)
@pytest.mark.parametrize("K", [64, 1000, 4096])
def test_flat_weights_most_nodes_are_adopted(K, flags):
    """The opposite of real data's weight degeneracy: all-gap rows (every site likelihood is 1, weights differ only through the
    priors), so most particles are adopted and most nodes of a rank event are written (lazy nodes: nearly every node is
    materialised, and many merges read a node written one launch earlier).  Bit-exact against the oracle, nodes included."""
    N, S = 9, 333
    g = np.ones((N, S, 4))
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, K, Q)
    for seed in (2, 11):
        out = ctx.sweep(seed, flags=flags)
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed, want_nodes=(K <= 64))
        check(out, ref, "flat K=%d seed %d" % (K, seed))
        assert len(np.unique(out['ancestors'][0])) > K // 3        # the regime this test is about
        if K <= 64:
            for r in range(N - 1):
                for k in (0, K // 2, K - 1):
                    assert same_bits(ctx.sweep_node(r, k), ref['nodes'][r, k]), "node (%d,%d)" % (r, k)
    ctx.close()


@pytest.mark.parametrize("flags", FORMS, ids=FORM_IDS)
def test_generic_leaf_row_no_codes(flags):
    """One leaf row that is neither one-hot nor all-ones: no leaf codes, every child goes through the row loads."""
    g = load_dataset('primate_data')['genome'][:8, :300].copy()
    g[2, 11] = [0.25, 0.5, 0.125, 0.125]
    N = 8
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 3.0)
    ctx = ctx_for(g, 512, Q)
    ctx.set_model(Q, PI, lam, lam)
    for seed in (0, 1, 2):
        check(ctx.sweep(seed, flags=flags), CO.sweep(g, Q, PI, lam, lam, 512, seed), "generic leaves seed %d" % seed)
    ctx.close()


@pytest.mark.parametrize("dataset,K", [('hohna_data_3', 200), ('hohna_data_5', 333), ('hohna_data_8', 256)])
def test_ds_datasets_with_33_to_64_taxa(dataset, K):
    """DS3 (36 taxa), DS5 (50), DS8 (64): the bookkeeping takes one wave per particle (64 lanes), still in the launch that also
    writes the adopted nodes.  Bit-exact against the oracle; stored and dead nodes on demand."""
    g = load_dataset(dataset)['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ctx = ctx_for(g, K, Q)
    for seed in (0, 3):
        out = ctx.sweep(seed)
        assert out['stats']['n_launches'] <= 3 * (N - 1) + 2, "bookkeeping and adopted-node writes did not share a launch"
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed, want_nodes=(seed == 3))
        check(out, ref, "%s K=%d seed %d" % (dataset, K, seed))
    for (r, k) in [(0, 0), (N - 2, K - 1), (N // 2, K // 3), (1, int(out['ancestors'][1, 0]))]:
        assert same_bits(ctx.sweep_node(r, k), ref['nodes'][r, k]), "node (%d,%d)" % (r, k)
    ctx.close()
