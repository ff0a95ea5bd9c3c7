"""Independent cross-check of the gradient oracle: the sweep written with PyTorch tensor ops in the shape of the
reference's TensorFlow graph (vcsmc.py:332-451: K-fold replicated cores, tf.gather resampling of cores and of the
previous log-likelihoods, matrix exponential, reparameterised Exponential samples), differentiated by
torch.autograd on CPU, against oracle/cpu_grad.py's hand-written reverse pass.  Discrete choices (resampling
indices, pair picks) and the uniforms are taken from the oracle and held constant, as the TensorFlow ops that
produce them carry no gradient.  PyTorch is test plumbing here; nothing in the product imports it."""
import numpy as np
import pytest

from oracle import cpu_grad as G
from oracle import cpu_ref as O

torch = pytest.importorskip("torch")


def _torch_logZ(genome, y_q, y_s, a_l, a_r, K, st, raw_q=True):
    """log Z-hat as a differentiable function of the reference's four variables."""
    dt = torch.float64
    N, S, A = genome.shape
    R = N - 1
    e = torch.exp(y_q) * (1.0 - torch.eye(A, dtype=dt))                  # get_Q, vcsmc.py:138-148
    q_entry = e / e.sum(dim=1, keepdim=True)
    Q = q_entry - torch.diag(q_entry.sum(dim=1))
    pi = torch.softmax(y_s, dim=0)                                       # vcsmc.py:133-136
    lam_l, lam_r = torch.exp(a_l), torch.exp(a_r)                        # vcsmc.py:119-120
    core = torch.tensor(genome, dtype=dt).unsqueeze(0).repeat(K, 1, 1, 1)    # [K, n, S, A]
    cnt = torch.ones((K, N), dtype=torch.int64)
    lw_rows, ll_prev = [], None
    bl_hist, br_hist = [], []
    ar = torch.arange(K)
    for r in range(R):
        n = N - r
        if r > 0:
            idx = torch.tensor(st['anc'][r], dtype=torch.int64)
            core, cnt = core[idx], cnt[idx]                              # tf.gather, vcsmc.py:286-288
            ll_tilde = ll_prev[idx]                                      # vcsmc.py:322-323
        else:
            ll_tilde = torch.full((K,), float(np.log(1.0 / K)), dtype=dt)
        co = torch.tensor(st['co'][r], dtype=torch.int64)
        rem = torch.tensor(st['rem'][r], dtype=torch.int64)
        bl = -torch.log(torch.tensor(st['Ul'][r], dtype=dt)) / lam_l[r]  # Exponential(rate).sample(), vcsmc.py:353-356
        br = -torch.log(torch.tensor(st['Ur'][r], dtype=dt)) / lam_r[r]
        bl_hist.append(bl)
        br_hist.append(br)
        Pl = torch.linalg.matrix_exp(bl[:, None, None] * Q)              # vcsmc.py:181-184
        Pr = torch.linalg.matrix_exp(br[:, None, None] * Q)
        L, Rr = core[ar, co[:, 0]], core[ar, co[:, 1]]
        new = torch.matmul(L, Pl) * torch.matmul(Rr, Pr)                 # vcsmc.py:185-187
        core = torch.cat([core[ar[:, None], rem], new[:, None]], dim=1)
        cnt = torch.cat([cnt[ar[:, None], rem], (cnt[ar, co[:, 0]] + cnt[ar, co[:, 1]])[:, None]], dim=1)
        site = torch.log(torch.matmul(core, pi))                         # vcsmc.py:240-242
        fprior = torch.tensor(np.sum(-O.log_double_factorial(2 * np.maximum(cnt.numpy(), 2) - 3), axis=1), dtype=dt)
        ll = site.sum(dim=(1, 2)) + fprior
        blh, brh = torch.stack(bl_hist), torch.stack(br_hist)            # rows 0..r of the slot-attached history
        ll = ll + torch.sum(-lam_l[r] * blh + torch.log(lam_l[r]), dim=0) + torch.sum(-lam_r[r] * brh + torch.log(lam_r[r]), dim=0)
        v_minus = torch.tensor(O.overcounting_correct(cnt.numpy()).astype(np.float64), dtype=dt)
        q = 1.0 / O.ncr2(n)
        lw = ll - ll_tilde - (torch.log(lam_l[r]) - lam_l[r] * bl + torch.log(lam_r[r]) - lam_r[r] * br) \
            + torch.log(v_minus) - (q if raw_q else float(np.log(q)))    # vcsmc.py:390-392
        lw_rows.append(lw)
        ll_prev = ll
    lws = torch.stack(lw_rows)
    return torch.sum(torch.logsumexp(lws, dim=1) - float(np.log(K)))     # compute_log_ZSMC, vcsmc.py:270-277


@pytest.mark.parametrize("flags", [O.QUIRK_Q1_RAW_Q, 0])
def test_hand_written_reverse_pass_equals_autodiff(flags):
    rng = np.random.default_rng(4)
    N, S, K = 6, 30, 10
    codes = rng.integers(0, 5, size=(N, S))
    genome = np.zeros((N, S, 4))
    for a in range(4):
        genome[..., a] = (codes == a) | (codes == 4)
    y_q = rng.normal(size=(4, 4)) * 0.3
    np.fill_diagonal(y_q, 0.0)
    y_s = rng.normal(size=4) * 0.3
    a_l, a_r = rng.normal(size=N - 1) * 0.2 + 1.3, rng.normal(size=N - 1) * 0.2 + 1.3
    e = np.exp(y_q)
    np.fill_diagonal(e, 0.0)
    Q = e / e.sum(axis=1, keepdims=True)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    pi = (np.exp(y_s) / np.exp(y_s).sum())[None, :]
    g = G.sweep_grad(genome, Q, pi, np.exp(a_l), np.exp(a_r), K, seed=21, flags=flags)
    v = G.to_variables(Q, pi, np.exp(a_l), np.exp(a_r), g)

    t = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (y_q, y_s, a_l, a_r)]
    logZ = _torch_logZ(genome, *t, K, g['struct'], raw_q=bool(flags & O.QUIRK_Q1_RAW_Q))
    assert abs(float(logZ.detach()) - g["logZ"]) < 1e-9 * abs(g["logZ"])
    logZ.backward()
    d_yq = t[0].grad.numpy().copy()
    np.fill_diagonal(d_yq, 0.0)                      # the diagonal of the variable is overwritten by set_diag (vcsmc.py:122)
    for mine, auto in ((v['d_y_q'], d_yq), (v['d_y_station'], t[1].grad.numpy()), (v['d_loglam_l'], t[2].grad.numpy()),
                       (v['d_loglam_r'], t[3].grad.numpy())):
        scale = np.max(np.abs(auto))
        assert np.max(np.abs(mine - auto)) < 1e-9 * scale, (mine, auto)


def _torch_logZ_twisted(genome, y_q, y_s, a_l, a_r, K, M, st):
    """The twisted sweep in the shape of vncsmc.py:295-416, 432-499: potentials of every (pair, sub-sample), normalised by
    logsumexp, the chosen one subtracted from the weight; nothing detached but the draws and the gather indices."""
    dt = torch.float64
    N, S, A = genome.shape
    R = N - 1
    e = torch.exp(y_q) * (1.0 - torch.eye(A, dtype=dt))
    q_entry = e / e.sum(dim=1, keepdim=True)
    Q = q_entry - torch.diag(q_entry.sum(dim=1))
    pi = torch.softmax(y_s, dim=0)
    lam_l, lam_r = torch.exp(a_l), torch.exp(a_r)
    core = torch.tensor(genome, dtype=dt).unsqueeze(0).repeat(K, 1, 1, 1)
    cnt = torch.ones((K, N), dtype=torch.int64)
    lw_rows, ll_prev = [], None
    bl_hist, br_hist = [], []
    ar = torch.arange(K)

    def ldf(c):
        return torch.tensor(O.log_double_factorial(2 * np.maximum(c.numpy(), 2) - 3), dtype=dt)

    def post(data_KxSxA, c_K):                                           # broadcast_compute_tree_posterior_K, vncsmc.py:217-233
        return torch.log(torch.matmul(data_KxSxA, pi)).sum(dim=1) - ldf(c_K)

    for r in range(R):
        n = N - r
        if r > 0:
            idx = torch.tensor(st['anc'][r], dtype=torch.int64)
            core, cnt = core[idx], cnt[idx]
            ll_tilde = ll_prev[idx]
        else:
            ll_tilde = torch.full((K,), float(np.log(1.0 / K)), dtype=dt)
        pairs = O.pair_list(n)
        b_l = -torch.log(torch.tensor(st['Ul'][r], dtype=dt)) / lam_l[r]    # [K, J], vncsmc.py:351-356
        b_r = -torch.log(torch.tensor(st['Ur'][r], dtype=dt)) / lam_r[r]
        cols = []
        for t_, (r1, r2) in enumerate(pairs):                                # vncsmc.py:341-374
            l_data, r_data = core[:, r1], core[:, r2]
            base = post(l_data, cnt[:, r1]) + post(r_data, cnt[:, r2])
            for m in range(M):
                j = t_ * M + m
                Pl = torch.linalg.matrix_exp(b_l[:, j, None, None] * Q)
                Pr = torch.linalg.matrix_exp(b_r[:, j, None, None] * Q)
                mtx = torch.matmul(l_data, Pl) * torch.matmul(r_data, Pr)
                cols.append(post(mtx, cnt[:, r1] + cnt[:, r2]) - base)
        pot = torch.stack(cols, dim=1)
        pot = pot - torch.logsumexp(pot, dim=1, keepdim=True)                # vncsmc.py:399-401
        js = torch.tensor(st['js'][r], dtype=torch.int64)
        logq = pot[ar, js]                                                   # vncsmc.py:315-316
        bl, br = b_l[ar, js], b_r[ar, js]                                    # :317-320
        bl_hist.append(bl)
        br_hist.append(br)
        ts = (st['js'][r] // M)
        co = torch.tensor(np.array([pairs[t_] for t_ in ts]), dtype=torch.int64)
        rem = torch.tensor(np.array([[i for i in range(n - 1, -1, -1) if i not in pairs[t_]] for t_ in ts]).reshape(K, n - 2),
                           dtype=torch.int64)
        Pl = torch.linalg.matrix_exp(bl[:, None, None] * Q)
        Pr = torch.linalg.matrix_exp(br[:, None, None] * Q)
        new = torch.matmul(core[ar, co[:, 0]], Pl) * torch.matmul(core[ar, co[:, 1]], Pr)
        core = torch.cat([core[ar[:, None], rem], new[:, None]], dim=1)
        cnt = torch.cat([cnt[ar[:, None], rem], (cnt[ar, co[:, 0]] + cnt[ar, co[:, 1]])[:, None]], dim=1)
        site = torch.log(torch.matmul(core, pi))
        fprior = torch.tensor(np.sum(-O.log_double_factorial(2 * np.maximum(cnt.numpy(), 2) - 3), axis=1), dtype=dt)
        ll = site.sum(dim=(1, 2)) + fprior
        blh, brh = torch.stack(bl_hist), torch.stack(br_hist)
        ll = ll + torch.sum(-lam_l[r] * blh + torch.log(lam_l[r]), dim=0) + torch.sum(-lam_r[r] * brh + torch.log(lam_r[r]), dim=0)
        v_minus = torch.tensor(O.overcounting_correct(cnt.numpy()).astype(np.float64), dtype=dt)
        lw = ll - ll_tilde - (torch.log(lam_l[r]) - lam_l[r] * bl + torch.log(lam_r[r]) - lam_r[r] * br) \
            + torch.log(v_minus) - logq                                      # vncsmc.py:489-491
        lw_rows.append(lw)
        ll_prev = ll
    lws = torch.stack(lw_rows)
    return torch.sum(torch.logsumexp(lws, dim=1) - float(np.log(K)))


@pytest.mark.parametrize("M", [1, 2])
def test_twisted_reverse_pass_equals_autodiff(M):
    rng = np.random.default_rng(6)
    N, S, K = 5, 20, 8
    codes = rng.integers(0, 5, size=(N, S))
    genome = np.zeros((N, S, 4))
    for a in range(4):
        genome[..., a] = (codes == a) | (codes == 4)
    y_q = rng.normal(size=(4, 4)) * 0.3
    np.fill_diagonal(y_q, 0.0)
    y_s = rng.normal(size=4) * 0.3
    a_l, a_r = rng.normal(size=N - 1) * 0.2 + 1.3, rng.normal(size=N - 1) * 0.2 + 1.3
    e = np.exp(y_q)
    np.fill_diagonal(e, 0.0)
    Q = e / e.sum(axis=1, keepdims=True)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    pi = (np.exp(y_s) / np.exp(y_s).sum())[None, :]
    g = G.sweep_grad_twisted(genome, Q, pi, np.exp(a_l), np.exp(a_r), K, M, seed=21)
    v = G.to_variables(Q, pi, np.exp(a_l), np.exp(a_r), g)
    t = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (y_q, y_s, a_l, a_r)]
    logZ = _torch_logZ_twisted(genome, *t, K, M, g['struct'])
    assert abs(float(logZ.detach()) - g["logZ"]) < 1e-9 * abs(g["logZ"])
    logZ.backward()
    d_yq = t[0].grad.numpy().copy()
    np.fill_diagonal(d_yq, 0.0)
    for mine, auto in ((v['d_y_q'], d_yq), (v['d_y_station'], t[1].grad.numpy()), (v['d_loglam_l'], t[2].grad.numpy()),
                       (v['d_loglam_r'], t[3].grad.numpy())):
        scale = np.max(np.abs(auto))
        assert np.max(np.abs(mine - auto)) < 1e-9 * scale, (mine, auto)
