"""The VI training step of VCSMC.train (vcsmc.py:488-491, 532-536): one minibatch sweep, its gradient, one
optimiser update of the reference's variables.

The reference builds `optimizer.minimize(self.cost)` over four TensorFlow variables and lets autodiff walk
the sweep.  Here the sweep and its reverse pass run on the device (phylo_sweep with PHYLO_KEEP_GRAPH, then
phylo_sweep_backward), which returns d logZ / d(lam_l, lam_r, pi, Q); this module owns what is left:
  * the variables and their parameterisation (vcsmc.py:119-148): log-rates, y_q, y_station;
  * the chain rules from the raw quantities to those variables (42 numbers, host NumPy);
  * tf.train.GradientDescentOptimizer / tf.train.AdamOptimizer update rules (TF 1.15 defaults);
  * the minibatch loop over site slices, including its skipped last slice (vcsmc.py:533, quirk Q9).
"""
from __future__ import annotations

import numpy as np

from . import _ffi, model


class Variables:
    """tf.trainable_variables() of the reference graph: 'left_branches_param', 'right_branches_param' (the
    exponents), 'Qmatrix' (y_q) and 'Stationary_probs' (y_station); the last two are constants under --jcmodel."""

    def __init__(self, N, branch_prior, jcmodel, A=4):
        self.jc = bool(jcmodel)
        self.a_l = np.zeros(N - 1) + branch_prior               # vcsmc.py:119
        self.a_r = np.zeros(N - 1) + branch_prior               # vcsmc.py:120
        self.y_q = model.init_y_q(A)                            # vcsmc.py:122
        self.y_station = np.zeros(A) + 1 / A                    # vcsmc.py:124

    def names(self):
        return ('a_l', 'a_r') if self.jc else ('a_l', 'a_r', 'y_q', 'y_station')

    def pack(self):
        """a_l | a_r | y_q | y_station, the layout of phylo_vi_gradients / phylo_vi_apply"""
        return np.concatenate([self.a_l, self.a_r, np.asarray(self.y_q, dtype=np.float64).reshape(-1), self.y_station])

    def unpack(self, p):
        R = self.a_l.shape[0]
        self.a_l, self.a_r = p[:R].copy(), p[R:2 * R].copy()
        if not self.jc:
            self.y_q, self.y_station = p[2 * R:2 * R + 16].reshape(4, 4).copy(), p[2 * R + 16:2 * R + 20].copy()

    def unpack_grads(self, g):
        R = self.a_l.shape[0]
        out = {'a_l': g[:R], 'a_r': g[R:2 * R]}
        if not self.jc:
            out['y_q'], out['y_station'] = g[2 * R:2 * R + 16].reshape(4, 4), g[2 * R + 16:2 * R + 20]
        return out

    def evaluate(self):
        """(Q, pi[1,A], lam_l, lam_r) as the graph evaluates them."""
        Q = model.jc_Q(self.y_q.shape[0]) if self.jc else model.get_Q(self.y_q)
        return Q, model.get_stationary_probs(self.y_station), np.exp(self.a_l), np.exp(self.a_r)


def chain_rules(v, Q, pi_1xA, lam_l, lam_r, raw):
    """d logZ / d variables from d logZ / d(lam, pi, Q).
    rates: lam = exp(a).  pi = softmax(y_station) (vcsmc.py:133-136).  Off-diagonal Q_ij = exp(y_ij) / sum_{j' != i}
    exp(y_ij'), Q_ii = -sum_j Q_ij (vcsmc.py:138-148); diagonal entries of y_q receive no gradient (set_diag)."""
    g = {'a_l': raw['d_lam_l'] * lam_l, 'a_r': raw['d_lam_r'] * lam_r}
    if not v.jc:
        pi = pi_1xA[0]
        g['y_station'] = pi * (raw['d_pi'] - np.dot(pi, raw['d_pi']))
        q = np.array(Q, dtype=np.float64)
        np.fill_diagonal(q, 0.0)
        dq = raw['d_Q'] - np.diag(raw['d_Q'])[:, None]
        np.fill_diagonal(dq, 0.0)
        g['y_q'] = q * (dq - np.sum(q * dq, axis=1, keepdims=True))
    return g


def mean_of_samples(samples, ctx=None):
    """Mean of (logZ, grads) over independent particle systems: the samples of this process, and with ctx (a context that has
    joined the ranks' communicator) those of every rank -- one host all-gather of the ~45 numbers per sample, summed in (rank,
    sample) order on every rank: the same bits everywhere, so the ranks' variables never drift apart, and the same bits as one
    process taking all the samples itself."""
    names = sorted(samples[0][1])
    rows = np.stack([np.concatenate([[float(z)]] + [np.asarray(g[n], dtype=np.float64).reshape(-1) for n in names]) for z, g in samples])
    if ctx is not None:
        rows = ctx.comm_allgather_blob(rows)
        rows = rows.reshape(-1, rows.shape[-1])
    tot = rows[0].copy()
    for r in range(1, rows.shape[0]):
        tot = tot + rows[r]
    tot = tot / rows.shape[0]
    out, o = {}, 1
    for n in names:
        shape = np.shape(samples[0][1][n])
        size = int(np.prod(shape)) if shape else 1
        out[n] = tot[o:o + size].reshape(shape)
        o += size
    return float(tot[0]), out


class GradientDescent:
    """tf.train.GradientDescentOptimizer(lr).minimize(cost): var <- var - lr d cost/d var, cost = -logZ."""

    def __init__(self, learning_rate):
        self.lr = float(learning_rate)

    def __str__(self):
        return 'GradientDescentOptimizer(learning_rate=%g)' % self.lr

    def apply(self, v, grads_logZ):
        for name in v.names():
            setattr(v, name, getattr(v, name) + self.lr * grads_logZ[name])

    def apply_packed(self, v, packed_grads):
        """the same update by the library (phylo_vi_apply) on the packed variables"""
        p = v.pack()
        _ffi.vi_apply(v.a_l.shape[0] + 1, v.jc, p, packed_grads, 0, self.lr)
        v.unpack(p)


class Adam:
    """tf.train.AdamOptimizer (TF 1.15 defaults beta1 .9, beta2 .999, epsilon 1e-8):
    lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);  var <- var - lr_t m / (sqrt(v) + epsilon)."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def __str__(self):
        return 'AdamOptimizer(learning_rate=%g)' % self.lr

    def apply(self, v, grads_logZ):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for name in v.names():
            g = -grads_logZ[name]                                # gradient of the cost
            m = self.m.get(name, 0.0) * self.b1 + (1.0 - self.b1) * g
            s = self.v.get(name, 0.0) * self.b2 + (1.0 - self.b2) * g * g
            self.m[name], self.v[name] = m, s
            setattr(v, name, getattr(v, name) - lr_t * m / (np.sqrt(s) + self.eps))

    def apply_packed(self, v, packed_grads):
        """the same update by the library (phylo_vi_apply) on the packed variables; its state lives in self.state (an optimiser
        object is driven through ONE of the two forms)"""
        p = v.pack()
        if getattr(self, 'state', None) is None:
            self.state = {'t': 0, 'm': np.zeros_like(p), 'v': np.zeros_like(p)}
        _ffi.vi_apply(v.a_l.shape[0] + 1, v.jc, p, packed_grads, 1, self.lr, self.b1, self.b2, self.eps, self.state)
        self.t = self.state['t']
        v.unpack(p)


def make_optimizer(name, learning_rate):
    """runner.py:30-33: 'Adam' selects Adam, anything else plain gradient descent (vcsmc.py:488-491)."""
    return Adam(learning_rate) if name == 'Adam' else GradientDescent(learning_rate)


class Trainer:
    """One device context sized for a minibatch of sites; `step` = sweep + reverse pass + update."""

    def __init__(self, genome_NxSxA, K, variables, optimizer, batch_sites, device=0, flags=_ffi.FLAGS_DEFAULT, nested=False, M=1,
                 native=True):
        """nested: the twisted proposal of vncsmc.py with M sub-samples per pair; its look-ahead potentials are differentiated
        like everything else (vncsmc.py:379-416 has no stop_gradient).  native: the host half of a step (model from the
        variables, chain rules, optimiser update) runs in the library (phylo_vi_gradients / phylo_vi_apply) instead of the NumPy
        statements of this module (~60 small array operations, 65 us of a 0.9 ms step); same formulas."""
        self.native = bool(native)
        self.genome = np.asarray(genome_NxSxA, dtype=np.float64)
        self.v, self.opt = variables, optimizer
        self.flags = (flags | _ffi.KEEP_GRAPH) & ~_ffi.TWISTING
        if nested:
            self.flags |= _ffi.TWISTING
        self.M = int(M) if nested else 1
        N = self.genome.shape[0]
        self.ctx = _ffi.Context(K, N, int(batch_sites), device=device)
        self.last = None
        self._sites = None

    def close(self):
        self.ctx.close()

    def gradients(self, sites, seed):
        """Sweep over genome[:, sites] and its gradient w.r.t. the variables.  Returns (logZ, grads, raw)."""
        Q, pi, lam_l, lam_r = self.v.evaluate()
        sites = np.asarray(sites)
        if self._sites is None or not np.array_equal(sites, self._sites):   # the slice on the device is still this one
            self.ctx.set_leaves(self.genome[:, sites, :])
            self._sites = sites.copy()
        self.ctx.set_model(Q, pi, lam_l, lam_r, jc69_closed_form=self.v.jc)
        self.ctx.sweep_async(int(seed), self.flags, self.M)
        raw = self.ctx.sweep_backward()                     # queued right behind the sweep: no host round trip in between
        out = self.ctx.sweep_fetch(arrays=False)
        raw['forward_ms'] = out['stats']['sweep_ms']
        return out['logZ'], chain_rules(self.v, Q, pi, lam_l, lam_r, raw), raw

    def _gradients_native(self, sites, seed):
        """gradients() through phylo_vi_gradients: (logZ, grads dict, raw timings, grads packed)"""
        sites = np.asarray(sites)
        if self._sites is None or not np.array_equal(sites, self._sites):
            self.ctx.set_leaves(self.genome[:, sites, :])
            self._sites = sites.copy()
        logZ, g, fwd, bwd = self.ctx.vi_gradients(int(seed), self.flags, self.M, self.v.jc, self.v.pack())
        raw = {'forward_ms': fwd.sweep_ms, 'backward_ms': bwd.sweep_ms, 'backward_host_ms': bwd.merge_ms,
               'backward_lists': 'device' if bwd.merge_launches else 'host', 'backward_launches': bwd.n_launches}
        return logZ, self.v.unpack_grads(g), raw, g

    def step(self, sites, seed, more_seeds=(), comm_ctx=None):
        """_, cost = sess.run([self.optimizer, self.cost], feed_dict={self.core: data_batch})  (vcsmc.py:534).
        Data-parallel training: more_seeds = further independent particle systems swept by this process for the same step,
        comm_ctx = a context that has joined the ranks (every rank sweeps its own systems with its own seeds); the optimiser takes
        ONE step on the mean gradient of all of them (mean_of_samples), the same step on every rank."""
        if self.native:
            logZ, grads, raw, packed = self._gradients_native(sites, seed)
        else:
            logZ, grads, raw = self.gradients(sites, seed)
        if more_seeds or comm_ctx is not None:
            samples = [(logZ, grads)]
            for s2 in more_seeds:
                z2, g2, raw = (self._gradients_native(sites, s2)[:3] if self.native else self.gradients(sites, s2))
                samples.append((z2, g2))
            logZ, grads = mean_of_samples(samples, comm_ctx)
            if self.native:
                packed = np.concatenate([np.asarray(grads[n], dtype=np.float64).reshape(-1) for n in ('a_l', 'a_r', 'y_q', 'y_station') if n in grads])
        if self.native:
            if packed.shape[0] < self.v.pack().shape[0]:     # (JC69: y_q and y_station are constants, their slots stay zero)
                packed = np.concatenate([packed, np.zeros(20)])
            self.opt.apply_packed(self.v, packed)
        else:
            self.opt.apply(self.v, grads)
        self.last = {'logZ': logZ, 'grads': grads, 'raw': raw}
        return -logZ
