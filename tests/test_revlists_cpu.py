"""The host side of the reverse pass (phylo_amd/csrc/phylo_revlists.h: adopters, parents, chunks, flagged nodes) against a plain
restatement in Python, on random genealogies from flat to degenerate.  No GPU: phylo_debug_reverse_lists runs the same functions
phylo_sweep_backward calls."""
import numpy as np
import pytest

from phylo_amd import _ffi

PCHUNK, HCHUNK, FREE = 8, 32, 1 << 30


def _random_genealogy(rng, N, K, survivors):
    """ancestors [N-2][K] with about `survivors` distinct values per rank event; children: a root of the adopted table, which after r
    rank events holds leaves and earlier nodes -- any leaf or any earlier node id is in range for the builders."""
    R = N - 1
    anc = np.zeros((max(R - 1, 0), K), dtype=np.int64)
    for r in range(R - 1):
        pool = rng.choice(K, size=min(survivors, K), replace=False)
        p = rng.dirichlet(np.full(len(pool), 0.3))
        anc[r] = rng.choice(pool, size=K, p=p)
    child = np.zeros((R, K, 2), dtype=np.int32)
    for r in range(R):
        for k in range(K):
            for side in range(2):
                if r == 0 or rng.random() < 0.5:
                    child[r, k, side] = rng.integers(0, N)
                else:                                        # a node of an earlier rank event, mostly one its ancestor line made
                    rp = rng.integers(0, r)
                    kp = anc[rp, k] if rp < R - 1 and rng.random() < 0.8 else rng.integers(0, K)
                    child[r, k, side] = N + rp * K + kp
    return anc, child


def _reference(N, K, anc, child, early_free, rows_form, lookahead):
    R = N - 1
    nn = R * K
    adopters = [[[] for _ in range(K)] for _ in range(R)]
    for r in range(1, R):
        for k in range(K):
            adopters[r][int(anc[r - 1, k])].append(k)
    parents = [[] for _ in range(nn)]
    flat = child.reshape(-1)
    for e in range(2 * nn):
        if flat[e] >= N:
            parents[flat[e] - N].append(e)
    flags = np.zeros(nn, dtype=np.int64)
    for x in lookahead:
        flags[x - N] |= 2
    slow_lists = []
    for r in range(R):
        ev = []
        for k in range(K):
            x = r * K + k
            if parents[x]:
                flags[x] |= 1
            if early_free and r + 1 < R and adopters[r + 1][k]:
                flags[x] |= 4
            if flags[x]:
                ev.append(x)
        slow_lists.append(ev)
    return adopters, parents, flags, slow_lists


@pytest.mark.parametrize("N,K,survivors,early,rows,seed", [
    (2, 1, 1, True, True, 0), (2, 5, 3, True, True, 1), (3, 4, 2, True, True, 2), (4, 3, 3, False, True, 3),
    (6, 64, 3, True, True, 4), (6, 64, 64, True, True, 5), (7, 130, 5, True, False, 6), (8, 257, 12, False, True, 7),
    (12, 512, 4, True, True, 8), (5, 1023, 2, True, True, 9),
])
def test_lists_match_restatement(N, K, survivors, early, rows, seed):
    rng = np.random.default_rng(seed)
    R = N - 1
    nn = R * K
    anc, child = _random_genealogy(rng, N, K, survivors)
    lookahead = [] if seed % 2 else [int(N + x) for x in rng.choice(nn, size=min(nn, 5), replace=False)]
    out = _ffi.debug_reverse_lists(N, K, anc, child, early, rows, lookahead)
    adopters, parents, flags, slow_lists = _reference(N, K, anc, child, early, rows, lookahead)
    # adopters, ascending, and the adopted particles by rank event
    adp_ref = []
    for r in range(1, R):
        off = out["ad_off"][r]
        assert off[0] == 0 and off[K] == K
        for k in range(K):
            assert list(out["ad_idx"][r][off[k]:off[k + 1]]) == adopters[r][k], (r, k)
    for r in range(R):
        ev = [r * K + k for k in range(K) if r + 1 < R and adopters[r + 1][k]]
        assert list(out["adp"][out["ev_adp0"][r]:out["ev_adp0"][r + 1]]) == ev, r
        adp_ref += ev
    assert out["n_adp"] == len(adp_ref)
    # parents; a parent without flags is marked free in the rows form
    po = out["par_off"]
    assert po[0] == 0 and po[nn] == out["n_par"] == sum(len(p) for p in parents)
    for x in range(nn):
        got = out["par_idx"][po[x]:po[x + 1]]
        if rows and early:   # entries of free parents first, ascending; those of flagged parents behind them, descending
            free = [e for e in parents[x] if flags[e >> 1] == 0]
            slow = [e for e in parents[x] if flags[e >> 1] != 0]
            want = free + slow[::-1]
        else:
            want = parents[x]
        assert list(got & (FREE - 1)) == want, x
        for e in got:
            assert bool(e & FREE) == (rows and flags[(int(e) & (FREE - 1)) >> 1] == 0)
    # flags, lists of flagged nodes by rank event, chunks of the heavy nodes numbered within the rank event
    ns = 0
    nch = 0
    for r in range(R):
        assert out["ev_slow0"][r] == ns and out["rank_chunk0"][r] == nch
        for x in slow_lists[r]:
            assert out["slow_idx"][ns] == x and out["slow_flag"][x] == (ns << 3 | flags[x])
            ns += 1
        for k in range(K):
            x = r * K + k
            if flags[x] == 0:
                assert out["slow_flag"][x] == 0
            n_par = len(parents[x])
            if n_par > PCHUNK:
                assert out["heavy"][x] == nch - out["rank_chunk0"][r]
                for b in range(0, n_par, HCHUNK):
                    assert out["chunk_beg"][nch] == po[x] + b and out["chunk_cnt"][nch] == min(HCHUNK, n_par - b)
                    nch += 1
            else:
                assert out["heavy"][x] == -1
    assert out["ev_slow0"][R] == ns == out["n_slow"] and out["rank_chunk0"][R] == nch == out["n_chunks"]
    per_event = np.diff(out["rank_chunk0"])
    assert out["max_chunks"] == (per_event.max() if len(per_event) else 0)


def test_bad_arguments_are_refused():
    with pytest.raises(_ffi.PhyloError):
        _ffi.debug_reverse_lists(1, 4, None, np.zeros((0, 4, 2), np.int32))
    with pytest.raises(_ffi.PhyloError):
        _ffi.debug_reverse_lists(3, 4, np.zeros((1, 4), np.int64), np.zeros((2, 4, 2), np.int32), lookahead_nodes=[1])   # a leaf id
