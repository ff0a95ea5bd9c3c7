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
        """csmc.py:237-257: two posets sampled uniformly (python's RNG, like the reference), fixed branch
        lengths 2, 2."""
        jump_chain_KxN[j, i + 1] = deepcopy(jump_chain_KxN[j, i])
        sample = random.sample(jump_chain_KxN[j, i][0], 2)
        q2 = 1 / self.ncr(len(jump_chain_KxN[j, i][0]), 2)
        particle1, particle2 = sample[0], sample[1]
        particle_coalesced = self.sort_string(particle1 + '+' + particle2)
        jump_chain_KxN[j, i + 1][0].remove(particle1)
        jump_chain_KxN[j, i + 1][0].remove(particle2)
        jump_chain_KxN[j, i + 1][0].append(particle_coalesced)
        bl1, bl2 = 2, 2
        return particle1, particle2, particle_coalesced, bl1, bl2, q2, jump_chain_KxN

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
        """csmc.py:335-349."""
        trees = [dic.keys() for dic in vertex_dicts]
        tree_probabilities = []
        for i in range(len(trees)):
            tree = trees[i]
            tree_probabilities.append(0)
            for k in range(K):
                if tree == trees[k]:
                    tree_probabilities[i] += weights_KxNm1[k, -1]
            tree_probabilities[i] /= K
        tree_probabilities /= 1 / K * sum(weights_KxNm1[:, -1])
        return list(tree_probabilities), trees

    def compute_norm(self, weights_KxNm1, K):
        """csmc.py:351-355."""
        norm = 1
        for i in range(1, self.n - 1):
            norm *= 1 / K * sum(weights_KxNm1[:, i])
        return norm

    def sample_phylogenies(self, K, resampling=False, showing=True):
        """csmc.py:357-454 with the per-root likelihoods evaluated on the GPU.  Returns
        (log_weights[K, n-1], tree_probabilities, norm, selected root Vertex)."""
        n = self.n
        jump_chain = [{} for i in range(n)]
        jump_chain[0][0] = self.taxa
        jump_chain_KxN = np.array([jump_chain] * K)
        log_weights_KxNm1 = np.zeros([K, n - 1])
        weights_KxNm1 = np.zeros([K, n - 1]) + 1
        log_likelihood = np.zeros([K, n - 1])
        log_likelihood_tilda = np.zeros(K) + 1
        vertex_dicts = [{} for k in range(K)]
        for j in range(K):
            for i in range(n):
                vertex_dicts[j][self.taxa[i]] = Vertex(id=self.taxa[i], data=self.genome_NxSxA[i])
        last_root = [None] * K
        for i in range(n - 1):
            if resampling and i > 0:
                jump_chain_KxN[:, i - 1] = self.resample(log_weights_KxNm1, jump_chain_KxN[:, i - 1], i - 1)
            for k in range(K):
                if i > 0:
                    log_likelihood_tilda[k] = 0
                    idx = random.randint(0, K - 1)
                    for key in vertex_dicts[idx]:
                        if vertex_dicts[idx][key].is_root:
                            log_likelihood_tilda[k] += self.compute_log_conditional_likelihood(vertex_dicts[idx][key])
                particle1, particle2, particle_coalesced, bl1, bl2, q, jump_chain_KxN = \
                    self.extend_partial_state(jump_chain_KxN, k, i)
                vertex_dicts[k][particle_coalesced] = Vertex(id=particle_coalesced, data=None)
                vertex_dicts[k][particle_coalesced].left = vertex_dicts[k][particle1]
                vertex_dicts[k][particle_coalesced].right = vertex_dicts[k][particle2]
                vertex_dicts[k][particle_coalesced].left_branch = bl1
                vertex_dicts[k][particle_coalesced].right_branch = bl2
                vertex_dicts[k][particle1].is_root = False
                vertex_dicts[k][particle2].is_root = False
                last_root[k] = vertex_dicts[k][particle_coalesced]
            for k in range(K):
                log_likelihood[k, i] = 0
                for key in vertex_dicts[k]:
                    if vertex_dicts[k][key].is_root:
                        log_likelihood[k, i] += self.compute_log_conditional_likelihood(vertex_dicts[k][key])
                v = self.overcounting_correct(vertex_dicts[k])
                if i > 0:
                    log_weights_KxNm1[k, i] = log_likelihood[k, i] - log_likelihood_tilda[k] + np.log(v) - np.log(q)
                    weights_KxNm1[k, i] = np.exp(log_weights_KxNm1[k, i])
            if showing:
                print('Computation in progress: step ' + str(i + 1))
        tree_probabilities, trees = self.get_tree_prob(vertex_dicts, weights_KxNm1, K)
        norm = self.compute_norm(weights_KxNm1, K)
        selected_idx = tree_probabilities.index(max(tree_probabilities))
        return log_weights_KxNm1, tree_probabilities, norm, last_root[selected_idx]
