"""Host-side logic that needs no GPU: CLI flags, dataset table, model matrices, the host RNG contract."""
import numpy as np
import pytest

import runner
from oracle import cpu_ref as O
from phylo_amd import datasets, model, rng
from phylo_amd.vcsmc import VCSMC, default_args, log_double_factorial, ncr


def test_cli_defaults_match_reference():
    a = runner.parse_args([])
    # runner.py:12-58
    assert a.dataset == 'primate_data' and a.n_particles == 10 and a.batch_size == 256
    assert a.learning_rate == 0.001 and a.num_epoch == 100 and a.optimizer == 'GradientDescentOptimizer'
    assert a.branch_prior == pytest.approx(np.log(10)) and a.M == 10
    assert a.nested is False and a.jcmodel is False and a.memory_optimization == 'on'
    assert runner.parse_args(['--jcmodel', 'True']).jcmodel is True
    assert runner.parse_args(['--jcmodel', 'yes']).jcmodel is False          # str(x).lower() == 'true'
    assert runner.parse_args(['--twisting', 'true']).nested is True           # README flag


def test_dataset_table():
    d = datasets.load_dataset('primate_data')
    assert d['genome'].shape == (12, 898, 4) and d['taxa'][:2] == ['S0', 'S1']
    assert int((d['genome'].sum(axis=2) == 4).sum()) == 30                     # 30 gap cells -> [1,1,1,1]
    d = datasets.load_dataset('primate_data_wang')
    assert d['genome'].shape == (9, 738, 4) and (d['genome'].sum(axis=2) == 1).all()
    assert datasets.load_dataset('hohna_data_1')['genome'].shape == (27, 1949, 4)
    assert datasets.load_dataset('load_strings')['genome'].shape == (4, 10, 4)
    assert datasets.load_dataset('simulate_data')['genome'].shape == (3, 5, 4)
    with pytest.raises(KeyError):
        datasets.load_dataset('hohna_data_7')                                  # contains 'N' (SURVEY F8): the reference's behaviour
    d7 = datasets.load_dataset('hohna_data_7', ambiguity='iupac')              # opt-in: 'N' -> all-ones row, like a gap
    assert d7['genome'].shape == (59, 1824, 4) and int((d7['genome'].sum(axis=2) == 4).sum()) == 28
    np.testing.assert_array_equal(datasets.form_dataset_from_strings(['RYN'], datasets.Alphabet_dir_iupac)['genome'][0],
                                  [[1, 0, 1, 0], [0, 1, 0, 1], [1, 1, 1, 1]])
    np.testing.assert_array_equal(datasets.load_dataset('hohna_data_1', ambiguity='iupac')['genome'],
                                  datasets.load_dataset('hohna_data_1')['genome'])
    with pytest.raises(FileNotFoundError):
        datasets.load_dataset('corona_data')
    with pytest.raises(ValueError):
        datasets.load_dataset('__import__("os").system("true")')               # no exec of dataset names
    s = datasets.synthetic_alignment(128, 500)
    assert s['genome'].shape == (128, 500, 4) and (s['genome'].sum(axis=2) == 1).all()
    np.testing.assert_array_equal(datasets.form_dataset_from_strings(['AC-T'], datasets.Alphabet_dir_blank)['genome'],
                                  O.form_dataset_from_strings(['AC-T'], O.ALPHABET_DIR_BLANK)['genome'])


def test_model_matches_oracle():
    rs = np.random.default_rng(0)
    y = rs.normal(size=(4, 4))
    np.testing.assert_array_equal(model.get_Q(y), O.get_Q(y))
    np.testing.assert_array_equal(model.jc_Q(), O.jc_Q())
    np.testing.assert_array_equal(model.get_stationary_probs(y[0]), O.get_stationary_probs(y[0]))
    np.testing.assert_allclose(model.branch_rates(12, np.log(10)), np.full(11, 10.0), rtol=1e-15)


def test_host_rng_contract_matches_oracle():
    x = rng.philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in x] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    for n in (2, 3, 12, 27, 130):
        a, b = rng.pair_order(40, n, 11, 2, k0=3), O.extend_partial_state(40, n, 11, 2, k0=3)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        both = np.concatenate([a[0], a[1]], axis=1)
        assert (np.sort(both, axis=1) == np.arange(n)).all()                   # a permutation of the slots


def test_vcsmc_constructor_mirrors_reference_attributes():
    d = datasets.load_dataset('load_strings')
    v = VCSMC(d, K=8, args=default_args(jcmodel=True))
    assert (v.K, v.N, v.S, v.A, v.M) == (8, 4, 10, 4, 10)
    np.testing.assert_array_equal(v.Qmatrix, model.jc_Q())
    np.testing.assert_allclose(v.left_branches_param, 10.0)
    v = VCSMC(d, K=8, args=default_args())
    np.testing.assert_allclose(v.Qmatrix, np.full((4, 4), 1 / 3) - np.eye(4) * 4 / 3, atol=1e-15)
    assert v.stationary_probs.shape == (1, 4)
    assert ncr(12, 2) == 66.0 and ncr(2, 2) == 1.0
    np.testing.assert_allclose(log_double_factorial(np.array([1, 3, 5, 7])), [0, np.log(3), np.log(15), np.log(105)])
    rec = np.array([[1, 1, 2, 4], [3, 1, 1, 1]])
    np.testing.assert_array_equal(v.overcounting_correct(rec), O.overcounting_correct(rec))


def test_adoption_from_draws_equals_adoption_from_indices():
    """What pk_rank_book_mat / pk_materialize_by_draws rely on: particle k is adopted iff some draw threshold
    thr = mulhi64(draw, total) lies in [cdf[k-1], cdf[k]) -- the same set as the distinct values of the index search
    (first i with cdf[i] > thr), also with zero weights (empty intervals), NaN / -inf weights and the all-equal case."""
    from oracle import cpu_ref as O
    rng = np.random.default_rng(5)
    cases = [rng.normal(size=257) * 30.0,                       # a few heavy particles, many underflowing to weight 0
             np.zeros(64),                                      # all equal
             np.where(rng.random(300) < 0.5, -np.inf, rng.normal(size=300)),
             np.full(40, -np.inf),                              # no finite weight: uniform integer weights
             np.r_[np.nan, rng.normal(size=99) * 5.0]]
    for logw in cases:
        for step in (1, 7):
            K = logw.shape[0]
            wi = O.resample_int_weights(logw)
            cdf = np.cumsum(wi, dtype=np.uint64)
            total = int(cdf[-1])
            x0, x1, _, _ = O.philox4x32(np.arange(K), step, O.STREAM_RESAMPLE, 0, 1234)
            draws = (x1.astype(np.uint64) << np.uint64(32)) | x0.astype(np.uint64)
            thr = O.mulhi64(draws, total)
            lo = np.r_[np.uint64(0), cdf[:-1]]
            adopted = np.array([bool(np.any((thr >= lo[k]) & (thr < cdf[k]))) for k in range(K)])
            idx = O.resample_indices(logw, 1234, step)
            assert np.array_equal(np.flatnonzero(adopted), np.unique(idx))
