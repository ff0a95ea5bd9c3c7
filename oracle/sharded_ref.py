"""Sharded restatement of the sweep (TEST INFRASTRUCTURE): the multi-GPU protocol of DESIGN.md in NumPy.

Each rank owns the particles [rank*K/G, (rank+1)*K/G).  Float state (node partial likelihoods) exists only on
the owner.  Integer state (root tables, leaf counts), two forms:
  * local_tables=True (the GPU default when sharded): a rank advances only ITS particles' rows; at a resampling
    the rows of an adopted ancestor are read from the ancestor's OWNER (`comm.gather_tables`, on the GPU a read
    of the owner's table slab over the peer mapping);
  * local_tables=False (PHYLO_REPLICATED_BOOK=1): rows of all K particles are REPLICATED and advanced on every
    rank from the shared counter-based draws.
lazy=True (GPU: from S >= 8192, or PHYLO_LAZY_NODES=1): a node is written only when its creator is adopted at the next
resampling.  Every rank derives all K resampling indices (the weights are replicated), so each owner knows which of its
nodes of the previous rank event were adopted; it writes them, a barrier orders the writes, then everybody merges.  A
read of a node that was never written is an error in this model.  Per rank event the ranks all-gather three K-vectors
(log-weights, log-likelihoods, node log-likelihoods); a child node owned by another rank is fetched from its
owner when (and only when) it is merged -- once per sweep and rank: the row is kept in a local cache (the GPU path's
pk_cache_claim / pk_cache_fill; cache_slots bounds it, a node that finds it full is fetched again at every merge, i.e. read in
place).  `comm` needs all_gather(np.ndarray) -> list of arrays and
fetch_node(owner, key) (tests wire these to torch.distributed gloo).  The result must equal cpu_ref.sweep.
"""
from __future__ import annotations

import numpy as np

from . import cpu_ref as O


def sweep_sharded(comm, rank, world, genome, Q, pi_1xA, lam_l, lam_r, K, seed, flags=O.QUIRK_Q1_RAW_Q, local_tables=False,
                  lazy=False, cache_slots=None):
    """lazy='draws': like lazy=True, but every owner finds ITS adopted nodes from the K draw thresholds (the GPU's
    pk_materialize_by_draws: k is adopted iff some mulhi64(draw, total) lies in [cdf[k-1], cdf[k])) instead of from the K indices."""
    N, S, A = genome.shape
    Kl = K // world
    k0 = rank * Kl
    local = slice(k0, k0 + Kl)
    roots = np.tile(np.arange(N), (K, 1))                    # node ids; leaves 0..N-1, node (r, k) = N + r*K + k
    cnt = np.ones((K, N), dtype=np.int64)
    leaf_ll = np.sum(np.log(np.matmul(genome, pi_1xA[0])), axis=1)
    rootll = np.tile(leaf_ll, (K, 1))
    pool = {}                                                # (r, k) -> [S,4] for LOCAL k only (written nodes)
    unwritten = {}                                           # lazy: nodes computed for their likelihood but not written
    fetched = 0
    cache, cache_full = {}, set()                            # remote nodes kept locally (None: unbounded; 0: none, always in place)
    log_weights, log_lik = np.zeros((N - 1, K)), np.zeros((N - 1, K))
    bls, brs = np.zeros((N - 1, Kl)), np.zeros((N - 1, Kl))
    ancestors = np.zeros((max(N - 2, 0), K), dtype=np.int64)
    ll_tilde = np.zeros(K) + np.log(1.0 / K)

    def node_data(node_id):
        nonlocal fetched
        if node_id < N:
            return genome[node_id]
        r_, k_ = divmod(node_id - N, K)
        owner = k_ // Kl
        if owner == rank:
            return pool[(r_, k_)]
        if (r_, k_) in cache:
            return cache[(r_, k_)]
        fetched += 1
        row = comm.fetch_node(owner, (r_, k_))               # over xGMI on the GPU path
        if (r_, k_) not in cache_full:
            if cache_slots is None or len(cache) < cache_slots:
                cache[(r_, k_)] = row                        # first merge of this node on this rank: its local copy from now on
            else:
                cache_full.add((r_, k_))                     # no room: this node stays remote for the rest of the sweep
        return row

    for r in range(N - 1):
        n = N - r
        if r > 0:
            idx = O.resample_indices(log_weights[r - 1], seed, r)        # identical on every rank
            if local_tables:                                            # ancestors' rows come from their owners
                parts = comm.gather_tables((roots[local], cnt[local], rootll[local]))
                roots = np.concatenate([p[0] for p in parts])
                cnt = np.concatenate([p[1] for p in parts])
                rootll = np.concatenate([p[2] for p in parts])
            if lazy:                                                     # owners write the nodes adopted just now
                if lazy == 'draws':
                    cdf = np.cumsum(O.resample_int_weights(log_weights[r - 1]), dtype=np.uint64)
                    x0, x1, _, _ = O.philox4x32(np.arange(K), r, O.STREAM_RESAMPLE, 0, seed)
                    thr = O.mulhi64((x1.astype(np.uint64) << np.uint64(32)) | x0.astype(np.uint64), int(cdf[-1]))
                    lo = np.r_[np.uint64(0), cdf[:-1]]
                    adopted = [k for k in range(k0, k0 + Kl) if np.any((thr >= lo[k]) & (thr < cdf[k]))]
                else:
                    adopted = sorted(set(int(v) for v in idx))
                for a_ in adopted:
                    if k0 <= a_ < k0 + Kl and (r - 1, a_) in unwritten:
                        pool[(r - 1, a_)] = unwritten.pop((r - 1, a_))
                unwritten.clear()                                        # the rest of that rank event's nodes are dead
                comm.barrier()
            roots, cnt, rootll = roots[idx], cnt[idx], rootll[idx]
            ll_tilde = log_lik[r - 1, idx]
            ancestors[r - 1] = idx
        co, rem, q = O.extend_partial_state(K, n, seed, r)               # replicated integer bookkeeping
        ar = np.arange(K)
        cl, cr = roots[ar, co[:, 0]], roots[ar, co[:, 1]]
        new_cnt = cnt[ar, co[:, 0]] + cnt[ar, co[:, 1]]
        roots_rem, cnt_rem, ll_rem = roots[ar[:, None], rem], cnt[ar[:, None], rem], rootll[ar[:, None], rem]
        bl, br = O.branch_samples(Kl, lam_l[r], lam_r[r], seed, r, k0=k0)
        bls[r], brs[r] = bl, br
        # serve the nodes other ranks need from me, then merge my own particles
        comm.serve_begin(pool)
        L = np.stack([node_data(int(c)) for c in cl[local]])
        R = np.stack([node_data(int(c)) for c in cr[local]])
        comm.serve_end()
        new = O.broadcast_conditional_likelihood_K(Q, L, R, bl, br)
        node_ll = np.sum(np.log(np.matmul(new, pi_1xA[0])), axis=1)
        for j in range(Kl):
            (unwritten if lazy else pool)[(r, k0 + j)] = new[j]
        cnt_new = np.concatenate([cnt_rem, new_cnt[:, None]], axis=1)
        fprior = np.sum(-O.log_double_factorial(2 * np.maximum(cnt_new, 2) - 3), axis=1)
        ll_r = ll_rem[local].sum(axis=1) + node_ll + fprior[local]
        ll_r = ll_r + np.sum(-lam_l[r] * bls[:r + 1] + np.log(lam_l[r]), axis=0) \
                    + np.sum(-lam_r[r] * brs[:r + 1] + np.log(lam_r[r]), axis=0)
        v_minus = O.overcounting_correct(cnt_new)[local]
        qterm = q if (flags & O.QUIRK_Q1_RAW_Q) else np.log(q)
        lw = ll_r - ll_tilde[local] - (np.log(lam_l[r]) - lam_l[r] * bl + np.log(lam_r[r]) - lam_r[r] * br) \
            + np.log(v_minus.astype(np.float64)) - qterm
        # the one collective of the rank event: three K-vectors
        g = comm.all_gather(np.stack([lw, ll_r, node_ll]))
        full = np.concatenate(g, axis=1)
        log_weights[r], log_lik[r], node_ll_all = full[0], full[1], full[2]
        new_ids = N + r * K + np.arange(K)
        roots = np.concatenate([roots_rem, new_ids[:, None]], axis=1)
        cnt = cnt_new
        rootll = np.concatenate([ll_rem, node_ll_all[:, None]], axis=1)
        if local_tables:                                                # rows of other ranks' particles are not mine to keep
            other = np.ones(K, dtype=bool)
            other[local] = False
            roots[other], cnt[other], rootll[other] = -1, -1, np.nan
    logZ = O.compute_log_ZSMC(np.concatenate([np.zeros((1, K)), log_weights]))
    return {'log_weights': log_weights, 'log_likelihood': log_lik, 'ancestors': ancestors, 'logZ': logZ,
            'remote_fetches': fetched, 'cached_nodes': len(cache), 'cache_overflow': len(cache_full)}
