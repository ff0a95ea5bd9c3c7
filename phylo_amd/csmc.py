"""CSMC: the reference's NumPy class surface (csmc.py:50-60, 129-454) over the MI355X library.

`Vertex`, `CSMC(datadict)`, `conditional_likelihood`, `compute_tree_likelihood`,
`compute_log_conditional_likelihood`, `resample`, `sample_phylogenies` keep the reference's names, argument
order and return structure.  Pruning and resampling run on the GPU (phylo_tree_loglik / phylo_resample).
Visualisation (networkx graphs, csmc.py:25-126, 175-215) is out of scope: the fourth return value of
sample_phylogenies is the selected tree's Vertex instead of a drawable Graph.
"""
from __future__ import annotations

import operator as op
import random
from copy import deepcopy
from functools import reduce

import numpy as np

from . import _ffi


class Vertex:
    """csmc.py:50-60."""

    def __init__(self, id=None, data=None):
        self.id = id
        self.data = data
        self.left = None
        self.right = None
        self.left_branch = None
        self.right_branch = None
        self.is_root = True
        self.data_done = False


class CSMC:
    """
    CSMC takes as input a dictionary (datadict) with two keys:
     taxa: a list of n strings denoting taxa
     genome_NxSxA: a 3 tensor of genomes for the n taxa one hot encoded
    """

    def __init__(self, datadict, device=0, seed=0):
        self.n = len(datadict['taxa'])
        self.taxa = datadict['taxa']
        self.genome_NxSxA = np.asarray(datadict['genome'], dtype=np.float64)
        self.s = len(self.genome_NxSxA[0])
        self.Qmatrix = np.array([[-1., .25, .5, .25],            # csmc.py:145-148
                                 [.25, -1., .25, .5],
                                 [.5, .25, -1., .25],
                                 [.25, .5, .25, -1.]]) / 10
        self.prior = np.ones(self.Qmatrix.shape[0]) / self.Qmatrix.shape[0]
        self.seed = seed
        self._draws = 0
        self._device = device
        self._ctx = None
        self._ctx_Q = None

    @property
    def Pmatrix(self):
        """csmc.py:149: expm(Q) (kept for attribute compatibility)."""
        return self._context().expm_batched(np.array([1.0]))[0]

    def _context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(max(self.n, 2), max(self.n, 2), self.s, device=self._device)
        if self._ctx_Q is None or not np.array_equal(self._ctx_Q, self.Qmatrix):    # Qmatrix is a plain attribute (csmc.py:553)
            N = self._ctx.N
            self._ctx.set_model(self.Qmatrix, self.prior, np.ones(N - 1), np.ones(N - 1))
            self._ctx_Q = np.array(self.Qmatrix, copy=True)
        return self._ctx

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def ncr(self, n, r):
        """csmc.py:154-159."""
        r = min(r, n - r)
        numer = reduce(op.mul, range(n, n - r, -1), 1)
        denom = reduce(op.mul, range(1, r + 1), 1)
        return numer / denom

    def resample(self, weights, jump_chain_K, i):
        """csmc.py:218-228: K draws with probabilities proportional to exp(weights[:, i]); returns
        jump_chain_K[indices].  The weights are normalised by their maximum first, so the real-data
        overflow of the reference (exp of log-weights ~ 600-1000 -> NaN) does not occur."""
        weights = np.asarray(weights, dtype=np.float64)
        self._draws += 1
        indices = self._context().resample(weights[:, i], self.seed, self._draws)
        return np.asarray(jump_chain_K)[indices]

    def sort_string(self, s):
        """csmc.py:231-235."""
        return '+'.join(sorted(s.split('+')))

    def extend_partial_state(self, jump_chain_KxN, j, i):
        """csmc.py:237-257: particle j's next jump-chain entry: two posets drawn uniformly without replacement
        (python's RNG, like the reference) are replaced by their sorted union; branch lengths are fixed at 2, 2;
        q2 = 1 / C(#posets, 2)."""
        current = jump_chain_KxN[j, i][0]
        first, second = random.sample(current, 2)
        merged = self.sort_string(first + '+' + second)
        nxt = deepcopy(jump_chain_KxN[j, i])
        nxt[0] = [p for p in nxt[0] if p != first and p != second] + [merged]
        jump_chain_KxN[j, i + 1] = nxt
        return first, second, merged, 2, 2, 1 / self.ncr(len(current), 2), jump_chain_KxN

    def conditional_likelihood(self, left, right, left_branch, right_branch):
        """csmc.py:300-309: (left.data @ expm(Q bl)) * (right.data @ expm(Q br)), [S,4]."""
        l = np.asarray(left.data, dtype=np.float64)[None]
        r = np.asarray(right.data, dtype=np.float64)[None]
        return self._context().cond_likelihood_K(l, r, np.array([left_branch], dtype=np.float64),
                                                 np.array([right_branch], dtype=np.float64))[0]

    def compute_tree_likelihood(self, prior, root):
        """csmc.py:311-316."""
        return np.dot(prior, np.asarray(root.data).T)

    def _flatten(self, v):
        """Vertex tree -> arrays for phylo_tree_loglik.  A vertex whose data is already known (a leaf, or
        an internal vertex with data_done, csmc.py:278-298) is a terminal row."""
        terminals, internals = [], []
        index = {}
        stack = [(v, False)]
        while stack:
            node, expanded = stack.pop()
            if id(node) in index:
                continue
            terminal = node.left is None or (node.data_done and node.data is not None)
            if terminal:
                index[id(node)] = ('t', len(terminals))
                terminals.append(node)
            elif expanded:
                index[id(node)] = ('i', len(internals))
                internals.append(node)
            else:
                stack.append((node, True))
                stack.append((node.right, False))
                stack.append((node.left, False))
        L = len(terminals)

        def num(node):
            kind, j = index[id(node)]
            return j if kind == 't' else L + j

        n_nodes = L + len(internals)
        left = np.full(n_nodes, -1, dtype=np.int32)
        right = np.full(n_nodes, -1, dtype=np.int32)
        bl, br = np.zeros(n_nodes), np.zeros(n_nodes)
        for j, node in enumerate(internals):
            left[L + j], right[L + j] = num(node.left), num(node.right)
            bl[L + j], br[L + j] = node.left_branch, node.right_branch
        leaves = np.stack([np.asarray(t.data, dtype=np.float64) for t in terminals])
        return left, right, bl, br, num(v), leaves, internals

    def compute_log_conditional_likelihood(self, v):
        """csmc.py:318-326: post-order pruning below v, then sum_s log(prior . v.data[s])."""
        left, right, bl, br, root, leaves, internals = self._flatten(v)
        loglik, root_data = self._context().tree_loglik(left, right, bl, br, root, leaves, self.prior)
        if internals:
            v.data = root_data
            v.data_done = True
        return loglik

    def overcounting_correct(self, vertex_dict):
        """csmc.py:328-333."""
        rho = 0
        for key in vertex_dict:
            if vertex_dict[key].is_root and vertex_dict[key].left is not None:
                rho += 1
        return 1 / rho

    def get_tree_prob(self, vertex_dicts, weights_KxNm1, K):
        """csmc.py:335-349: particle i's probability = mean last-rank weight of the particles holding the same vertex
        set, divided by the mean last-rank weight (sums run left to right over k, like the reference's loops)."""
        trees = [d.keys() for d in vertex_dicts]
        last = [float(x) for x in np.asarray(weights_KxNm1)[:, -1]]
        with np.errstate(all='ignore'):
            mean_all = np.float64(1 / K) * np.float64(sum(last))
            probs = []
            for t in trees:
                acc = 0.0
                for k in range(K):
                    if trees[k] == t:
                        acc += last[k]
                probs.append(float(np.float64(acc / K) / mean_all))
        return probs, trees

    def compute_norm(self, weights_KxNm1, K):
        """csmc.py:351-355: product over ranks 1..n-2 of the mean weight."""
        w = np.asarray(weights_KxNm1)
        norm = np.float64(1.0)
        with np.errstate(all='ignore'):
            for i in range(1, self.n - 1):
                norm = norm * (np.float64(1 / K) * np.float64(sum(float(x) for x in w[:, i])))
        return float(norm)

    def sample_phylogenies(self, K, resampling=False, showing=True):
        """csmc.py:357-454 with every per-root likelihood evaluated on the GPU.  Same control flow and quirks
        (weights of rank 0 stay 0; log_likelihood_tilda of a particle is the forest likelihood of a
        RANDOM particle of the partially updated step; resampling permutes jump chains only).  Returns
        (log_weights[K, n-1], tree_probabilities, norm, root Vertex of the most probable tree)."""
        n = self.n
        chain0 = [{} for _ in range(n)]
        chain0[0][0] = self.taxa
        jump_chain_KxN = np.array([chain0] * K)
        log_w = np.zeros((K, n - 1))
        w = np.ones((K, n - 1))
        forests = [{t: Vertex(id=t, data=self.genome_NxSxA[i]) for i, t in enumerate(self.taxa)} for _ in range(K)]
        newest = [None] * K

        def forest_loglik(forest):
            return sum(self.compute_log_conditional_likelihood(v) for v in forest.values() if v.is_root)

        for i in range(n - 1):
            if resampling and i > 0:
                jump_chain_KxN[:, i - 1] = self.resample(log_w, jump_chain_KxN[:, i - 1], i - 1)
            ll_tilda = np.ones(K)
            qs = np.zeros(K)
            for k in range(K):
                if i > 0:
                    ll_tilda[k] = forest_loglik(forests[random.randint(0, K - 1)])
                p1, p2, merged, bl1, bl2, qs[k], jump_chain_KxN = self.extend_partial_state(jump_chain_KxN, k, i)
                v = Vertex(id=merged, data=None)
                v.left, v.right, v.left_branch, v.right_branch = forests[k][p1], forests[k][p2], bl1, bl2
                forests[k][p1].is_root = forests[k][p2].is_root = False
                forests[k][merged] = newest[k] = v
            if i > 0:
                for k in range(K):
                    log_w[k, i] = forest_loglik(forests[k]) - ll_tilda[k] + np.log(self.overcounting_correct(forests[k])) \
                        - np.log(qs[k])
                    with np.errstate(over='ignore'):
                        w[k, i] = np.exp(log_w[k, i])      # overflows on real data, as in the reference
            if showing:
                print('Computation in progress: step ' + str(i + 1))
        tree_probabilities, trees = self.get_tree_prob(forests, w, K)
        norm = self.compute_norm(w, K)
        best = tree_probabilities.index(max(tree_probabilities))
        return log_w, tree_probabilities, norm, newest[best]
