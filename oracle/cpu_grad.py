"""CPU oracle (NumPy) for the VI training step: the gradient of log Z-hat (TEST INFRASTRUCTURE ONLY).

The reference trains by TensorFlow autodiff of cost = -log Z-hat through the tf.while_loop of the sweep
(vcsmc.py:445-447, 488-491, 534).  What that derivative is, read off the graph:
  * discrete choices are constants: resampling indices (tf.random.categorical, vcsmc.py:285), the pair pick
    (tf.nn.top_k indices, :304-305) and every tf.gather index;
  * branch lengths are reparameterised samples b = -log(U)/rate of tfp Exponential (vcsmc.py:353-356), so the
    gradient flows through them to the rates (pathwise);
  * everything else (expm, the merges, log, logsumexp, softmax/exp parameterisations) is differentiated.
This module restates that derivative by hand in reverse mode over a node pool and integer root tables, and
`finite_difference` checks it against central differences of the forward pass with the discrete structure and
the uniforms frozen (tests/test_oracle_grad.py).  scipy.linalg.expm_frechet supplies the derivative of expm
with respect to Q (the adjoint identity <Ybar, L(A,E)> = <L(A^T, Ybar), E>).

Returned gradients are with respect to the RAW quantities lam_l[r], lam_r[r], pi[a], Q[i][j]; chain rules
to the reference's variables (log-rates, softmax logits, row-softmax logits) are in phylo_amd/train.py and
mirrored in `to_variables` here.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm, expm_frechet

from . import cpu_ref as O


def forward(genome, Q, pi_1xA, lam_l, lam_r, K, seed, flags=O.QUIRK_Q1_RAW_Q, struct=None):
    """The sweep of cpu_ref.sweep on a node pool.  `struct` (from a previous call) freezes the discrete choices
    and the uniforms, so that the result is a smooth function of (Q, pi, lam_l, lam_r)."""
    N, S, A = genome.shape
    R = N - 1
    pi = pi_1xA[0]
    draw = struct is None
    if draw:
        struct = {'anc': [None] * R, 'co': [None] * R, 'rem': [None] * R, 'Ul': np.zeros((R, K)), 'Ur': np.zeros((R, K))}
    nodes = {}                                              # (r, k) -> [S,4]
    roots = np.tile(np.arange(N), (K, 1))                   # node ids: leaves 0..N-1, (r,k) -> N + r*K + k
    cnt = np.ones((K, N), dtype=np.int64)
    leaf_ll = np.sum(np.log(genome @ pi), axis=1)
    rootll = np.tile(leaf_ll, (K, 1))
    lw, ll = np.zeros((R, K)), np.zeros((R, K))
    bl, br = np.zeros((R, K)), np.zeros((R, K))
    Pl, Pr = np.zeros((R, K, 4, 4)), np.zeros((R, K, 4, 4))
    child = np.zeros((R, K, 2), dtype=np.int64)
    tables = []                                             # post-merge root tables per rank
    lt = np.zeros(K) + np.log(1.0 / K)
    ar = np.arange(K)

    def data(node_id):
        return genome[node_id] if node_id < N else nodes[divmod(node_id - N, K)]

    for r in range(R):
        n = N - r
        if r > 0:
            idx = O.resample_indices(lw[r - 1], seed, r) if draw else struct['anc'][r]
            struct['anc'][r] = idx
            roots, cnt, rootll = roots[idx], cnt[idx], rootll[idx]
            lt = ll[r - 1, idx]
        if draw:
            co, rem, _ = O.extend_partial_state(K, n, seed, r)
            struct['co'][r], struct['rem'][r] = co, rem
            x0, x1, x2, x3 = O.philox4x32(ar, r, O.STREAM_BRANCH, 0, seed)
            struct['Ul'][r], struct['Ur'][r] = O.u64_to_unit_open_closed(x0, x1), O.u64_to_unit_open_closed(x2, x3)
        co, rem = struct['co'][r], struct['rem'][r]
        q = 1.0 / O.ncr2(n)
        bl[r], br[r] = -np.log(struct['Ul'][r]) / lam_l[r], -np.log(struct['Ur'][r]) / lam_r[r]
        cl, cr = roots[ar, co[:, 0]], roots[ar, co[:, 1]]
        child[r, :, 0], child[r, :, 1] = cl, cr
        node_ll = np.zeros(K)
        for k in range(K):
            Pl[r, k], Pr[r, k] = expm(Q * bl[r, k]), expm(Q * br[r, k])
            X = (data(int(cl[k])) @ Pl[r, k]) * (data(int(cr[k])) @ Pr[r, k])
            nodes[(r, k)] = X
            node_ll[k] = np.sum(np.log(X @ pi))
        new_cnt = cnt[ar, co[:, 0]] + cnt[ar, co[:, 1]]
        roots = np.concatenate([roots[ar[:, None], rem], (N + r * K + ar)[:, None]], axis=1)
        cnt = np.concatenate([cnt[ar[:, None], rem], new_cnt[:, None]], axis=1)
        rootll = np.concatenate([rootll[ar[:, None], rem], node_ll[:, None]], axis=1)
        tables.append(roots.copy())
        fprior = np.sum(-O.log_double_factorial(2 * np.maximum(cnt, 2) - 3), axis=1)
        ll[r] = rootll.sum(axis=1) + fprior \
            + np.sum(-lam_l[r] * bl[:r + 1] + np.log(lam_l[r]), axis=0) + np.sum(-lam_r[r] * br[:r + 1] + np.log(lam_r[r]), axis=0)
        v_minus = O.overcounting_correct(cnt)
        qterm = q if (flags & O.QUIRK_Q1_RAW_Q) else np.log(q)
        lw[r] = ll[r] - lt - (np.log(lam_l[r]) - lam_l[r] * bl[r] + np.log(lam_r[r]) - lam_r[r] * br[r]) \
            + np.log(v_minus.astype(np.float64)) - qterm
    logZ = O.compute_log_ZSMC(np.concatenate([np.zeros((1, K)), lw]))
    return {'logZ': logZ, 'lw': lw, 'll': ll, 'bl': bl, 'br': br, 'Pl': Pl, 'Pr': Pr, 'child': child, 'nodes': nodes,
            'tables': tables, 'struct': struct}


def sweep_grad(genome, Q, pi_1xA, lam_l, lam_r, K, seed, flags=O.QUIRK_Q1_RAW_Q, struct=None):
    """Forward + reverse: d logZ / d(lam_l, lam_r, pi, Q)."""
    N, S, A = genome.shape
    R = N - 1
    pi = pi_1xA[0]
    f = forward(genome, Q, pi_1xA, lam_l, lam_r, K, seed, flags, struct)
    st, lw, bl, br, child, nodes, tables = f['struct'], f['lw'], f['bl'], f['br'], f['child'], f['nodes'], f['tables']
    # d logZ / d lw_r[k]: softmax over particles
    om = np.exp(lw - lw.max(axis=1, keepdims=True))
    om /= om.sum(axis=1, keepdims=True)
    # d logZ / d ll_r[k]: its own weight, minus the weights of the particles that adopt it (ll_tilde, vcsmc.py:322-323)
    G = om.copy()
    for r in range(R - 1):
        np.subtract.at(G[r], st['anc'][r + 1], om[r + 1])
    # coefficient of sum_s log(pi . X_x[s]) for every root slot, propagated down the adoption chains
    C = [None] * R
    for r in range(R - 1, -1, -1):
        n1 = N - r - 1                                       # entries of the post-merge table
        C[r] = np.repeat(G[r][:, None], n1, axis=1).copy()
        if r + 1 < R:
            rem = st['rem'][r + 1]                           # [K, n1 - 2]: slot of the adopted table kept at position p'
            for kp in range(K):
                a = st['anc'][r + 1][kp]
                for pp in range(n1 - 2):
                    C[r][a, rem[kp, pp]] += C[r + 1][kp, pp]
    alpha = {(r, k): C[r][k, N - r - 2] for r in range(R) for k in range(K)}
    # reverse sweep over the nodes, newest first
    Xbar = {}
    d_pi = np.zeros(4)
    for k in range(K):                                       # leaves still are roots of tables: pi . leaf[s] depends on pi
        for p in range(N - 1):
            x = int(tables[0][k, p])
            if x < N:
                d_pi += C[0][k, p] * np.sum(genome[x] / (genome[x] @ pi)[:, None], axis=0)
    Pl_bar, Pr_bar = np.zeros((R, K, 4, 4)), np.zeros((R, K, 4, 4))
    parents = {}                                             # node id -> list of (r, k, side)
    for r in range(R):
        for k in range(K):
            for side in (0, 1):
                c = int(child[r, k, side])
                if c >= N:
                    parents.setdefault(c, []).append((r, k, side))

    def data(node_id):
        return genome[node_id] if node_id < N else nodes[divmod(node_id - N, K)]

    for r in range(R - 1, -1, -1):
        for k in range(K):
            X = nodes[(r, k)]
            lik = X @ pi
            xb = alpha[(r, k)] * pi[None, :] / lik[:, None]
            d_pi += alpha[(r, k)] * np.sum(X / lik[:, None], axis=0)
            for (rp, kp, side) in parents.get(N + r * K + k, []):
                L, Rr = data(int(child[rp, kp, 0])), data(int(child[rp, kp, 1]))
                if side == 0:
                    xb = xb + (Xbar[(rp, kp)] * (Rr @ f['Pr'][rp, kp])) @ f['Pl'][rp, kp].T
                else:
                    xb = xb + (Xbar[(rp, kp)] * (L @ f['Pl'][rp, kp])) @ f['Pr'][rp, kp].T
            Xbar[(r, k)] = xb
            L, Rr = data(int(child[r, k, 0])), data(int(child[r, k, 1]))
            u, v = L @ f['Pl'][r, k], Rr @ f['Pr'][r, k]
            Pl_bar[r, k] = L.T @ (xb * v)
            Pr_bar[r, k] = Rr.T @ (xb * u)
    # transition matrices -> branch lengths and Q
    d_Q = np.zeros((4, 4))
    bl_bar, br_bar = np.zeros((R, K)), np.zeros((R, K))
    for r in range(R):
        for k in range(K):
            bl_bar[r, k] = np.sum(Pl_bar[r, k] * (Q @ f['Pl'][r, k]))
            br_bar[r, k] = np.sum(Pr_bar[r, k] * (Q @ f['Pr'][r, k]))
            d_Q += bl[r, k] * expm_frechet((Q * bl[r, k]).T, Pl_bar[r, k], compute_expm=False)
            d_Q += br[r, k] * expm_frechet((Q * br[r, k]).T, Pr_bar[r, k], compute_expm=False)
    # explicit occurrences of the branch lengths and of the rates in ll_r and in the proposal term
    d_lam_l, d_lam_r = np.zeros(R), np.zeros(R)
    for j in range(R):
        bl_bar[j] += -np.sum(G[j:] * lam_l[j:, None], axis=0) + om[j] * lam_l[j]
        br_bar[j] += -np.sum(G[j:] * lam_r[j:, None], axis=0) + om[j] * lam_r[j]
    for r in range(R):
        d_lam_l[r] = np.sum(G[r] * np.sum(-bl[:r + 1] + 1.0 / lam_l[r], axis=0) - om[r] * (1.0 / lam_l[r] - bl[r]))
        d_lam_r[r] = np.sum(G[r] * np.sum(-br[:r + 1] + 1.0 / lam_r[r], axis=0) - om[r] * (1.0 / lam_r[r] - br[r]))
        d_lam_l[r] += np.sum(bl_bar[r] * (-bl[r] / lam_l[r]))           # b = -log(U)/lambda
        d_lam_r[r] += np.sum(br_bar[r] * (-br[r] / lam_r[r]))
    f.update({'d_lam_l': d_lam_l, 'd_lam_r': d_lam_r, 'd_pi': d_pi, 'd_Q': d_Q, 'omega': om, 'G': G, 'alpha': alpha,
              'Pl_bar': Pl_bar, 'Pr_bar': Pr_bar, 'Xbar': Xbar})
    return f


def finite_difference(genome, Q, pi_1xA, lam_l, lam_r, K, seed, struct, which, index, h=1e-6, flags=O.QUIRK_Q1_RAW_Q):
    """Central difference of logZ with the discrete structure and the uniforms frozen."""
    def val(delta):
        Q2, pi2, ll2, lr2 = Q.copy(), pi_1xA.copy(), lam_l.copy(), lam_r.copy()
        {'Q': Q2, 'pi': pi2[0], 'lam_l': ll2, 'lam_r': lr2}[which][index] += delta
        return forward(genome, Q2, pi2, ll2, lr2, K, seed, flags, struct)['logZ']
    return (val(h) - val(-h)) / (2 * h)


def to_variables(Q, pi_1xA, lam_l, lam_r, g):
    """Chain rules to the reference's variables: log-rates (vcsmc.py:119-120), y_station (softmax, :133-136),
    y_q (row-softmax over the off-diagonal, diagonal = -row sum, :138-148; diagonal entries of the variable get no
    gradient because of tf.linalg.set_diag at :122)."""
    pi = pi_1xA[0]
    d_ystation = pi * (g['d_pi'] - np.dot(pi, g['d_pi']))
    q = Q.copy()
    np.fill_diagonal(q, 0.0)
    dq = g['d_Q'] - np.diag(g['d_Q'])[:, None]              # Q_ii = -sum_j q_ij
    np.fill_diagonal(dq, 0.0)
    d_yq = q * (dq - np.sum(q * dq, axis=1, keepdims=True))
    return {'d_loglam_l': g['d_lam_l'] * lam_l, 'd_loglam_r': g['d_lam_r'] * lam_r, 'd_y_station': d_ystation, 'd_y_q': d_yq}


# ------------------------------------------------------------------------------------------------------------------------
# The twisted / nested proposal (vncsmc.py:295-416, 432-499): what TensorFlow autodiff of cost = -log Z-hat differentiates there.
#   * the look-ahead potentials pot[k, j] of EVERY (pair, sub-sample) enter the weight through the normalised log-potential of
#     the chosen one, potentials - logsumexp (vncsmc.py:399-401, 315-316, 491): d logZ / d pot[k, j] = omega_r[k] (softmax_j -
#     [j = chosen]); the categorical draw itself (tf.random.categorical, :298) and every gather index are constants;
#   * each potential is post(merged) - post(left) - post(right) (vncsmc.py:362-366): it depends on pi, on the two roots' partial
#     likelihoods (so the roots of the ADOPTED table receive adjoints from every pair they take part in, at every rank event they
#     are alive) and, through expm(Q b) with b = -log(U)/lambda_r reparameterised (:351-356), on Q and on this rank's rates;
#   * the rest of the graph is the plain sweep's with the chosen pair / chosen sub-sample's branch lengths, and the remaining roots
#     kept in DESCENDING slot order (:305).
# ------------------------------------------------------------------------------------------------------------------------
def _twist_uniforms(K, n_pairs, M, seed, r):
    j = np.arange(n_pairs * M)[None, :]
    x0, x1, x2, x3 = O.philox4x32(np.arange(K)[:, None], r, O.STREAM_TWIST, j, seed)
    return O.u64_to_unit_open_closed(x0, x1), O.u64_to_unit_open_closed(x2, x3)


def forward_twisted(genome, Q, pi_1xA, lam_l, lam_r, K, M, seed, struct=None):
    """cpu_ref.sweep_twisted on a node pool; `struct` freezes the discrete choices and the uniforms."""
    N, S, A = genome.shape
    R = N - 1
    pi = pi_1xA[0]
    draw = struct is None
    if draw:
        struct = {'anc': [None] * R, 'js': [None] * R, 'Ul': [None] * R, 'Ur': [None] * R}
    nodes = {}
    roots = np.tile(np.arange(N), (K, 1))
    cnt = np.ones((K, N), dtype=np.int64)
    lw, ll = np.zeros((R, K)), np.zeros((R, K))
    bl, br = np.zeros((R, K)), np.zeros((R, K))
    lt = np.zeros(K) + np.log(1.0 / K)
    ar = np.arange(K)
    rec = {'ad_roots': [], 'ad_cnt': [], 'pot': [], 'b_l': [], 'b_r': [], 'pairs': [], 'co': [], 'rem': [], 'tables': []}
    child = np.zeros((R, K, 2), dtype=np.int64)
    Pl, Pr = np.zeros((R, K, 4, 4)), np.zeros((R, K, 4, 4))

    def data(node_id):
        return genome[node_id] if node_id < N else nodes[divmod(node_id - N, K)]

    def rowll(node_id):
        return np.sum(np.log(data(node_id) @ pi))

    def ldf(c):
        return float(O.log_double_factorial(np.array([2 * max(int(c), 2) - 3]))[0])

    for r in range(R):
        n = N - r
        if r > 0:
            idx = O.resample_indices(lw[r - 1], seed, r) if draw else struct['anc'][r]
            struct['anc'][r] = idx
            roots, cnt = roots[idx], cnt[idx]
            lt = ll[r - 1, idx]
        pairs = O.pair_list(n)
        if draw:
            struct['Ul'][r], struct['Ur'][r] = _twist_uniforms(K, len(pairs), M, seed, r)
        b_l = -np.log(struct['Ul'][r]) / lam_l[r]
        b_r = -np.log(struct['Ur'][r]) / lam_r[r]
        pot = np.zeros((K, len(pairs) * M))
        for k in range(K):
            for t, (r1, r2) in enumerate(pairs):
                X1, X2 = data(int(roots[k, r1])), data(int(roots[k, r2]))
                base = (rowll(int(roots[k, r1])) - ldf(cnt[k, r1])) + (rowll(int(roots[k, r2])) - ldf(cnt[k, r2]))
                for m in range(M):
                    j = t * M + m
                    Y = (X1 @ expm(Q * b_l[k, j])) * (X2 @ expm(Q * b_r[k, j]))
                    pot[k, j] = np.sum(np.log(Y @ pi)) - ldf(cnt[k, r1] + cnt[k, r2]) - base
        if draw:
            struct['js'][r], _ = O.twist_draw(pot, seed, r)
        js = struct['js'][r]
        mx = pot.max(axis=1, keepdims=True)
        logq = pot[ar, js] - (mx[:, 0] + np.log(np.sum(np.exp(pot - mx), axis=1)))
        co = np.array([pairs[t] for t in js // M], dtype=np.int64)
        rem = np.array([[i for i in range(n - 1, -1, -1) if i not in pairs[t]] for t in js // M], dtype=np.int64).reshape(K, n - 2)
        rec['ad_roots'].append(roots.copy()); rec['ad_cnt'].append(cnt.copy()); rec['pot'].append(pot)
        rec['b_l'].append(b_l); rec['b_r'].append(b_r); rec['pairs'].append(pairs); rec['co'].append(co); rec['rem'].append(rem)
        bl[r], br[r] = b_l[ar, js], b_r[ar, js]
        cl, cr = roots[ar, co[:, 0]], roots[ar, co[:, 1]]
        child[r, :, 0], child[r, :, 1] = cl, cr
        for k in range(K):
            Pl[r, k], Pr[r, k] = expm(Q * bl[r, k]), expm(Q * br[r, k])
            nodes[(r, k)] = (data(int(cl[k])) @ Pl[r, k]) * (data(int(cr[k])) @ Pr[r, k])
        new_cnt = cnt[ar, co[:, 0]] + cnt[ar, co[:, 1]]
        roots = np.concatenate([roots[ar[:, None], rem], (N + r * K + ar)[:, None]], axis=1)
        cnt = np.concatenate([cnt[ar[:, None], rem], new_cnt[:, None]], axis=1)
        rec['tables'].append(roots.copy())
        rootll = np.array([[rowll(int(x)) for x in roots[k]] for k in range(K)])
        fprior = np.sum(-O.log_double_factorial(2 * np.maximum(cnt, 2) - 3), axis=1)
        ll[r] = rootll.sum(axis=1) + fprior \
            + np.sum(-lam_l[r] * bl[:r + 1] + np.log(lam_l[r]), axis=0) + np.sum(-lam_r[r] * br[:r + 1] + np.log(lam_r[r]), axis=0)
        v_minus = O.overcounting_correct(cnt)
        lw[r] = ll[r] - lt - (np.log(lam_l[r]) - lam_l[r] * bl[r] + np.log(lam_r[r]) - lam_r[r] * br[r]) \
            + np.log(v_minus.astype(np.float64)) - logq
    logZ = O.compute_log_ZSMC(np.concatenate([np.zeros((1, K)), lw]))
    out = {'logZ': logZ, 'lw': lw, 'll': ll, 'bl': bl, 'br': br, 'Pl': Pl, 'Pr': Pr, 'child': child, 'nodes': nodes, 'struct': struct, 'M': M}
    out.update(rec)
    return out


def sweep_grad_twisted(genome, Q, pi_1xA, lam_l, lam_r, K, M, seed, struct=None):
    """Forward + reverse of the twisted sweep: d logZ / d(lam_l, lam_r, pi, Q)."""
    N, S, A = genome.shape
    R = N - 1
    pi = pi_1xA[0]
    f = forward_twisted(genome, Q, pi_1xA, lam_l, lam_r, K, M, seed, struct)
    st, lw, bl, br, child, nodes, tables = f['struct'], f['lw'], f['bl'], f['br'], f['child'], f['nodes'], f['tables']

    def data(node_id):
        return genome[node_id] if node_id < N else nodes[divmod(node_id - N, K)]

    om = np.exp(lw - lw.max(axis=1, keepdims=True))
    om /= om.sum(axis=1, keepdims=True)
    G = om.copy()
    for r in range(R - 1):
        np.subtract.at(G[r], st['anc'][r + 1], om[r + 1])
    # tau[r][k, j] = d logZ / d pot_r[k, j]
    tau = []
    for r in range(R):
        pot, js = f['pot'][r], st['js'][r]
        sg = np.exp(pot - pot.max(axis=1, keepdims=True))
        sg /= sg.sum(axis=1, keepdims=True)
        t = om[r][:, None] * sg
        t[np.arange(K), js] -= om[r]
        tau.append(t)
    # coefficient of sum_s log(pi . X) per root slot of the POST-merge table of rank event r (plain part), plus, per slot of the
    # ADOPTED table of rank event r, the potentials' -post(left) - post(right) terms: ctw[r][k, x] = -sum_{j contains x} tau
    ctw = []
    for r in range(R):
        n = N - r
        c = np.zeros((K, n))
        for t, (r1, r2) in enumerate(f['pairs'][r]):
            tj = tau[r][:, t * M:(t + 1) * M].sum(axis=1)
            c[:, r1] -= tj
            c[:, r2] -= tj
        ctw.append(c)
    C = [None] * R
    for r in range(R - 1, -1, -1):
        n1 = N - r - 1
        C[r] = np.repeat(G[r][:, None], n1, axis=1).copy()
        if r + 1 < R:
            rem = f['rem'][r + 1]
            for kp in range(K):
                a = st['anc'][r + 1][kp]
                for pp in range(n1 - 2):
                    C[r][a, rem[kp, pp]] += C[r + 1][kp, pp]
                C[r][a, :] += ctw[r + 1][kp]                 # every slot of the table particle kp adopted (= a's post-merge table)
    alpha = {(r, k): C[r][k, N - r - 2] for r in range(R) for k in range(K)}
    d_pi = np.zeros(4)
    d_Q = np.zeros((4, 4))
    d_lam_l, d_lam_r = np.zeros(R), np.zeros(R)
    # leaves as roots: pi . leaf[s] depends on pi.  Post-merge tables of rank event 0 (plain part) and the adopted table of
    # rank event 0 (the leaves themselves, twist part)
    leafpi = np.array([np.sum(genome[x] / (genome[x] @ pi)[:, None], axis=0) for x in range(N)])
    for k in range(K):
        for p in range(N - 1):
            x = int(tables[0][k, p])
            if x < N:
                d_pi += C[0][k, p] * leafpi[x]
        for x in range(N):
            d_pi += ctw[0][k, x] * leafpi[x]
    # the potentials' merged rows: adjoints of the two roots, of the transition matrices and of pi, newest rank event first
    Xextra = {}                                              # node id -> [S,4] adjoint from look-ahead merges
    for r in range(R - 1, -1, -1):
        ad = f['ad_roots'][r]
        for k in range(K):
            for t, (r1, r2) in enumerate(f['pairs'][r]):
                n1_, n2_ = int(ad[k, r1]), int(ad[k, r2])
                X1, X2 = data(n1_), data(n2_)
                for m in range(M):
                    j = t * M + m
                    tj = tau[r][k, j]
                    if tj == 0.0:
                        continue
                    b1, b2 = f['b_l'][r][k, j], f['b_r'][r][k, j]
                    P1, P2 = expm(Q * b1), expm(Q * b2)
                    u, v = X1 @ P1, X2 @ P2
                    Y = u * v
                    lik = Y @ pi
                    Yb = tj * pi[None, :] / lik[:, None]
                    d_pi += tj * np.sum(Y / lik[:, None], axis=0)
                    P1b, P2b = X1.T @ (Yb * v), X2.T @ (Yb * u)
                    if n1_ >= N:
                        Xextra[n1_] = Xextra.get(n1_, 0.0) + (Yb * v) @ P1.T
                    if n2_ >= N:
                        Xextra[n2_] = Xextra.get(n2_, 0.0) + (Yb * u) @ P2.T
                    b1b, b2b = np.sum(P1b * (Q @ P1)), np.sum(P2b * (Q @ P2))
                    d_lam_l[r] += b1b * (-b1 / lam_l[r])
                    d_lam_r[r] += b2b * (-b2 / lam_r[r])
                    d_Q += b1 * expm_frechet((Q * b1).T, P1b, compute_expm=False)
                    d_Q += b2 * expm_frechet((Q * b2).T, P2b, compute_expm=False)
    # reverse sweep over the nodes, newest first (the plain part with the extra adjoints)
    Xbar = {}
    Pl_bar, Pr_bar = np.zeros((R, K, 4, 4)), np.zeros((R, K, 4, 4))
    parents = {}
    for r in range(R):
        for k in range(K):
            for side in (0, 1):
                c = int(child[r, k, side])
                if c >= N:
                    parents.setdefault(c, []).append((r, k, side))
    for r in range(R - 1, -1, -1):
        for k in range(K):
            X = nodes[(r, k)]
            lik = X @ pi
            xb = alpha[(r, k)] * pi[None, :] / lik[:, None]
            d_pi += alpha[(r, k)] * np.sum(X / lik[:, None], axis=0)
            if (N + r * K + k) in Xextra:
                xb = xb + Xextra[N + r * K + k]
            for (rp, kp, side) in parents.get(N + r * K + k, []):
                L, Rr = data(int(child[rp, kp, 0])), data(int(child[rp, kp, 1]))
                if side == 0:
                    xb = xb + (Xbar[(rp, kp)] * (Rr @ f['Pr'][rp, kp])) @ f['Pl'][rp, kp].T
                else:
                    xb = xb + (Xbar[(rp, kp)] * (L @ f['Pl'][rp, kp])) @ f['Pr'][rp, kp].T
            Xbar[(r, k)] = xb
            L, Rr = data(int(child[r, k, 0])), data(int(child[r, k, 1]))
            u, v = L @ f['Pl'][r, k], Rr @ f['Pr'][r, k]
            Pl_bar[r, k] = L.T @ (xb * v)
            Pr_bar[r, k] = Rr.T @ (xb * u)
    bl_bar, br_bar = np.zeros((R, K)), np.zeros((R, K))
    for r in range(R):
        for k in range(K):
            bl_bar[r, k] = np.sum(Pl_bar[r, k] * (Q @ f['Pl'][r, k]))
            br_bar[r, k] = np.sum(Pr_bar[r, k] * (Q @ f['Pr'][r, k]))
            d_Q += bl[r, k] * expm_frechet((Q * bl[r, k]).T, Pl_bar[r, k], compute_expm=False)
            d_Q += br[r, k] * expm_frechet((Q * br[r, k]).T, Pr_bar[r, k], compute_expm=False)
    for j in range(R):
        bl_bar[j] += -np.sum(G[j:] * lam_l[j:, None], axis=0) + om[j] * lam_l[j]
        br_bar[j] += -np.sum(G[j:] * lam_r[j:, None], axis=0) + om[j] * lam_r[j]
    for r in range(R):
        d_lam_l[r] += np.sum(G[r] * np.sum(-bl[:r + 1] + 1.0 / lam_l[r], axis=0) - om[r] * (1.0 / lam_l[r] - bl[r]))
        d_lam_r[r] += np.sum(G[r] * np.sum(-br[:r + 1] + 1.0 / lam_r[r], axis=0) - om[r] * (1.0 / lam_r[r] - br[r]))
        d_lam_l[r] += np.sum(bl_bar[r] * (-bl[r] / lam_l[r]))
        d_lam_r[r] += np.sum(br_bar[r] * (-br[r] / lam_r[r]))
    f.update({'d_lam_l': d_lam_l, 'd_lam_r': d_lam_r, 'd_pi': d_pi, 'd_Q': d_Q, 'omega': om, 'G': G, 'tau': tau, 'ctw': ctw})
    return f


def finite_difference_twisted(genome, Q, pi_1xA, lam_l, lam_r, K, M, seed, struct, which, index, h=1e-6):
    def val(delta):
        Q2, pi2, ll2, lr2 = Q.copy(), pi_1xA.copy(), lam_l.copy(), lam_r.copy()
        {'Q': Q2, 'pi': pi2[0], 'lam_l': ll2, 'lam_r': lr2}[which][index] += delta
        return forward_twisted(genome, Q2, pi2, ll2, lr2, K, M, seed, struct)['logZ']
    return (val(h) - val(-h)) / (2 * h)
