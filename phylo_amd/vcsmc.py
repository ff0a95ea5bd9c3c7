"""VCSMC: the reference's class surface (vcsmc.py:103-645) over the MI355X library.

Same constructor, attribute names and method names/argument meaning as the reference.  Where the reference
builds TensorFlow graph nodes, these methods take and return NumPy arrays and run the computation on the
GPU through the C ABI (phylo_amd/_ffi.py).  No TensorFlow, no PyTorch, no CPU fallback for the hot ops.

Differences that are deliberate:
  * randomness is the counter-based contract of DESIGN.md (`seed` attribute; the reference sets no seed);
  * jump chains are rebuilt on the host from integer merge records, with the correct per-particle gather
    (the reference's vcsmc.py:306-307 gathers without the row offset, fixed in vncsmc.py);
  * train() takes the reference's optimiser steps (vcsmc.py:488-491, 532-536) with the gradient of the device's
    reverse pass (phylo_amd/train.py) instead of TensorFlow autodiff; with args.nested that is the reverse pass of the
    twisted proposal (vncsmc.py:568-640), every look-ahead potential differentiated.
"""
from __future__ import annotations

from datetime import datetime
from types import SimpleNamespace

import numpy as np

from . import _ffi, model, rng
from . import train as train_mod


def ncr(n, r):
    """vcsmc.py:23-27: n choose r as product / product (true division -> float)."""
    numer = np.prod(np.arange(n - r + 1, n + 1), dtype=np.int64) if r > 0 else 1
    denom = np.prod(np.arange(1, r + 1), dtype=np.int64) if r > 0 else 1
    return numer / denom


def log_double_factorial(n):
    """vcsmc.py:40-57: log n!! via the loop n, n-2, ... while >= 2; works on arrays."""
    n = np.asarray(n, dtype=np.float64).copy()
    result = np.zeros_like(n)
    while np.count_nonzero(n >= 2):
        result = np.where(n >= 2, result + np.log(np.where(n >= 2, n, 1.0)), result)
        n = n - 2
    return result


def gather_across_2d(a, idx, a_shape_1=None, idx_shape_1=None):
    """vcsmc.py:60-77: [a[k][idx[k]] for k]."""
    a, idx = np.asarray(a), np.asarray(idx)
    return a[np.arange(a.shape[0])[:, None], idx]


def gather_across_core(a, idx, a_shape_1=None, idx_shape_1=None, A=4):
    """vcsmc.py:80-97: per-particle row gather of a [K,N,S,A] core."""
    a, idx = np.asarray(a), np.asarray(idx)
    return a[np.arange(a.shape[0])[:, None], idx]


def default_args(**kw):
    """The reference's argparse defaults (runner.py:12-58) as a namespace."""
    d = dict(dataset='primate_data', n_particles=10, batch_size=256, learning_rate=0.001, num_epoch=100,
             optimizer='GradientDescentOptimizer', branch_prior=np.log(10), M=10, nested=False, jcmodel=False,
             memory_optimization='on', seed=0, n_gpus=1)
    d.update(kw)
    return SimpleNamespace(**d)


class VCSMC:
    """
    VCSMC takes as input a dictionary (datadict) with two keys:
     taxa: a list of n strings denoting taxa
     genome_NxSxA: a 3 tensor of genomes for the n taxa one hot encoded
    """

    def __init__(self, datadict, K, args=None, device=0):
        self.args = args if args is not None else default_args()
        self.taxa = datadict['taxa']
        self.genome_NxSxA = np.asarray(datadict['genome'], dtype=np.float64)
        self.K = K
        self.M = self.args.M
        self.N = len(self.genome_NxSxA)
        self.S = len(self.genome_NxSxA[0])
        self.A = len(self.genome_NxSxA[0, 0])
        self.seed = int(getattr(self.args, 'seed', 0) or 0)
        # the trainable variables (vcsmc.py:119-130) and what the graph derives from them
        self.variables = train_mod.Variables(self.N, self.args.branch_prior, self.args.jcmodel, self.A)
        self._sync_from_variables()
        self._device = device
        self._ctx = None
        self._sweeps = 0

    def _sync_from_variables(self):
        v = self.variables
        self.left_branches_param = np.exp(v.a_l)                                         # vcsmc.py:119
        self.right_branches_param = np.exp(v.a_r)                                        # vcsmc.py:120
        self.y_q, self.y_station = v.y_q, v.y_station
        self.Qmatrix = model.jc_Q(self.A) if v.jc else self.get_Q()                      # vcsmc.py:122-129
        self.stationary_probs = self.get_stationary_probs()

    # ---- device context -----------------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            n_gpus = int(getattr(self.args, 'n_gpus', 1) or 1)
            if n_gpus > 1:
                # one process per GPU (python -m torch.distributed.run --nproc-per-node N runner.py --n_gpus N ...):
                # the K particles are sharded over the ranks, resampling stays global
                import os
                from .rendezvous import exchange_comm_id
                world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
                if world != n_gpus:
                    raise RuntimeError("--n_gpus %d needs %d ranks (python -m torch.distributed.run --nproc-per-node %d ...), "
                                       "found WORLD_SIZE=%d" % (n_gpus, n_gpus, n_gpus, world))
                self._device = int(os.environ.get('LOCAL_RANK', '0')) % max(_ffi.device_count(), 1)
                self._rank, self._world = rank, world
            ctx = _ffi.Context(self.K, self.N, self.S, self.A, device=self._device)
            ctx.set_leaves(self.genome_NxSxA)
            if n_gpus > 1:
                cid = exchange_comm_id(self._rank, self._world, _ffi.comm_unique_id if self._rank == 0 else None)
                ctx.comm_init(self._rank, self._world, cid)
            self._ctx = ctx
        self._ctx.set_model(self.Qmatrix, self.stationary_probs, self.left_branches_param, self.right_branches_param,
                            jc69_closed_form=bool(self.args.jcmodel))
        return self._ctx

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    # ---- model (vcsmc.py:133-148) -------------------------------------------------------------------
    def get_stationary_probs(self):
        """ Compute stationary probabilities of the Q matrix """
        return model.get_stationary_probs(self.y_station)

    def get_Q(self):
        """Off-diagonal terms by the softmax function, diagonal = -row sum (vcsmc.py:138-148)."""
        return model.get_Q(self.y_q)

    # ---- per-rank ops -------------------------------------------------------------------------------
    def conditional_likelihood(self, l_data, r_data, l_branch, r_branch):
        """vcsmc.py:150-161: (l_data @ expm(Q l_branch)) * (r_data @ expm(Q r_branch)), [S,A]."""
        l = np.asarray(l_data, dtype=np.float64)[None]
        r = np.asarray(r_data, dtype=np.float64)[None]
        return self._context().cond_likelihood_K(l, r, np.array([l_branch], dtype=np.float64),
                                                 np.array([r_branch], dtype=np.float64))[0]

    def broadcast_conditional_likelihood_K(self, l_data_KxSxA, r_data_KxSxA, l_branch_samples_K, r_branch_samples_K):
        """vcsmc.py:180-188."""
        return self._context().cond_likelihood_K(l_data_KxSxA, r_data_KxSxA, l_branch_samples_K, r_branch_samples_K)

    def compute_forest_posterior(self, data_KxXxSxA, leafnode_num_record, r=None):
        """vcsmc.py:231-245 (r only fixes the reshape in the reference; shapes are taken from the data)."""
        return self._context().forest_loglik(data_KxXxSxA, leafnode_num_record)

    def overcounting_correct(self, leafnode_num_record):
        """vcsmc.py:247-252."""
        rec = np.asarray(leafnode_num_record)
        return np.sum(rec - (rec == 1).astype(np.int32), axis=1)

    def compute_log_ZSMC(self, log_weights):
        """vcsmc.py:270-277."""
        return self._context().log_zsmc(np.atleast_2d(log_weights))

    def resample(self, core, leafnode_num_record, JC_K, log_weights, step=1):
        """vcsmc.py:279-289: indices ~ Categorical(softmax(log_weights)), K draws; gathers core / record /
        jump chains by them.  `step` is the rank event the draw belongs to (RNG counter)."""
        indices = self._context().resample(log_weights, self.seed, step)
        return (np.asarray(core)[indices], np.asarray(leafnode_num_record)[indices], np.asarray(JC_K)[indices],
                indices)

    def extend_partial_state(self, JCK, r):
        """vcsmc.py:291-316: two root slots to coalesce per particle, the remaining slots, q = 1/C(N-r,2)
        and the new jump-chain strings."""
        n = self.N - r
        q = 1 / ncr(n, 2)
        coalesced_indices, remaining_indices = rng.pair_order(self.K, n, self.seed, r)
        JCK = np.asarray(JCK, dtype=object)
        keep = gather_across_2d(JCK, remaining_indices)
        particles = gather_across_2d(JCK, coalesced_indices)
        coalesced = np.array([a + '+' + b for a, b in particles], dtype=object)
        return coalesced_indices, remaining_indices, q, np.concatenate([keep, coalesced[:, None]], axis=1)

    def get_log_likelihood(self, log_likelihood):
        """vcsmc.py:254-268, including its use of the LEFT rates for the right multiplier (quirk Q8)."""
        l_exponent = self.left_branches.T * self.left_branches_param[None, :]
        r_exponent = self.right_branches.T * self.right_branches_param[None, :]
        l_multiplier = np.log(self.left_branches_param)[None, :]
        r_multiplier = np.log(self.left_branches_param)[None, :]
        left_branches_logprior = np.sum(l_multiplier - l_exponent, axis=1)
        right_branches_logprior = np.sum(r_multiplier - r_exponent, axis=1)
        return log_likelihood[self.N - 2] + log_double_factorial(2 * self.N - 3) - left_branches_logprior \
            - right_branches_logprior

    # ---- the sweep ----------------------------------------------------------------------------------
    def sample_phylogenies(self, seed=None, flags=None):
        """vcsmc.py:406-451 (vncsmc.py:505-560 with args.nested): the N-1 rank events on the device.  Sets
        the reference's attributes and returns the ELBO (log Z-hat)."""
        if flags is None:
            flags = _ffi.FLAGS_DEFAULT | (_ffi.TWISTING if getattr(self.args, 'nested', False) else 0)
        self._twisted = bool(flags & _ffi.TWISTING)
        seed = self.seed + self._sweeps if seed is None else int(seed)
        self._sweeps += 1
        ctx = self._context()
        out = ctx.sweep(seed, flags=flags, M=self.M)
        if ctx.K_local != ctx.K:                         # sharded: every rank assembles all K particles' outputs
            for key in ('log_weights', 'log_likelihood', 'left_branches', 'right_branches', 'ancestors'):
                out[key] = ctx.comm_allgather_columns(out[key])
            out['merges'] = np.ascontiguousarray(
                np.moveaxis(ctx.comm_allgather_columns(np.moveaxis(out['merges'], 1, 2)), 2, 1))
        self.log_weights = out['log_weights']            # rows 1..N-1 of the reference's tensor
        self.log_likelihood = out['log_likelihood']
        self.left_branches = out['left_branches']
        self.right_branches = out['right_branches']
        self.merges = out['merges']
        self.ancestors = out['ancestors']
        self.elbo = out['logZ']
        self.cost = -self.elbo
        self.log_likelihood_R = self.get_log_likelihood(self.log_likelihood)
        self.stats = out['stats']
        self._last_seed = seed
        anc = out['ancestors']
        self.log_likelihood_tilde = (out['log_likelihood'][self.N - 3, anc[self.N - 3]] if self.N > 2
                                     else np.zeros(self.K) + np.log(1 / self.K))
        self.jump_chain_tensor, self.v_minus = self._final_tables()
        return self.elbo

    def _final_tables(self):
        """Root-table strings and v_minus after the last rank event, replayed on the host from the
        integer records (ancestors, merges) and the pair-order contract."""
        K, N = self.K, self.N
        jc = np.array([list(self.taxa)] * K, dtype=object)
        rec = np.ones((K, N), dtype=np.int64)
        for r in range(N - 1):
            if r > 0:
                idx = self.ancestors[r - 1]
                jc, rec = jc[idx], rec[idx]
            if self._twisted:                      # chosen pair from the device; the rest in descending slot order
                co = self.merges[r]
                rem = np.array([[i for i in range(N - r - 1, -1, -1) if i != a and i != b] for a, b in co],
                               dtype=np.int64).reshape(K, N - r - 2)
            else:
                co, rem = rng.pair_order(K, N - r, self._last_seed, r)
                assert np.array_equal(co, self.merges[r]), "host replay of the pair pick disagrees with the device"
            new = np.array([a + '+' + b for a, b in gather_across_2d(jc, co)], dtype=object)
            jc = np.concatenate([gather_across_2d(jc, rem), new[:, None]], axis=1)
            rec = np.concatenate([gather_across_2d(rec, rem), gather_across_2d(rec, co).sum(axis=1)[:, None]], axis=1)
        return jc, self.overcounting_correct(rec)

    @property
    def jump_chains(self):
        return self.jump_chain_tensor

    def batch_slices(self, data, batch_size):
        """Site minibatches (vcsmc.py:453-464): S // batch_size draws of batch_size sites without replacement from the
        sites not used yet, then the leftover sites as a last slice.  Draws come from python's global RNG by
        random.sample over the unused-site list in CPython set-difference order, so a seeded `random` reproduces the
        reference's slices."""
        import random
        unused = list(range(data.shape[2]))
        out = []
        for _ in range(len(unused) // batch_size):
            picked = random.sample(unused, batch_size)
            out.append(picked)
            unused = list(set(unused) - set(picked))
        return out + [unused] if unused else out

    def newick(self, k=0):
        """Newick string of particle k's final tree, rebuilt from the integer merge records of the last
        sweep (branch lengths = the sampled left/right branches of each coalescence)."""
        K, N = self.K, self.N
        tab = [[str(t) for t in self.taxa] for _ in range(K)]
        for r in range(N - 1):
            if r > 0:
                idx = self.ancestors[r - 1]
                tab = [list(tab[i]) for i in idx]
            if self._twisted:
                rems = [[i for i in range(N - r - 1, -1, -1) if i != a and i != b] for a, b in self.merges[r]]
            else:
                rems = rng.pair_order(K, N - r, self._last_seed, r)[1]
            new = []
            for kk in range(K):
                a, b = self.merges[r][kk]
                node = '(%s:%.6g,%s:%.6g)' % (tab[kk][a], self.left_branches[r, kk], tab[kk][b], self.right_branches[r, kk])
                new.append([tab[kk][i] for i in rems[kk]] + [node])
            tab = new
        return tab[k][0] + ';'

    def _save_results(self, save_dir, initial, history):
        """run_parameters.txt and results.p with the reference's keys (vcsmc.py:503-516, 618-642); no plots."""
        import os
        import pickle
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "run_parameters.txt"), "w") as rp:
            rp.write('Initial evaluation of ELBO : ' + str(initial) + '\n')
            for key, v in vars(self.args).items():
                rp.write(str(key) + ' : ' + str(v) + '\n')
            rp.write(str(getattr(self, 'optimizer', '')))
        elbos = np.asarray(history['cost'])
        best = int(np.argmax(elbos)) if len(elbos) else 0
        resultDict = {'cost': elbos, 'nParticles': self.K, 'nTaxa': self.N, 'lr': self.lr,
                      'log_weights': np.asarray(history['log_weights']), 'Qmatrices': np.asarray(history['Qmatrices']),
                      'left_branches': history['left_branches'], 'right_branches': history['right_branches'],
                      'log_lik': np.asarray(history['log_lik']), 'll_tilde': np.asarray(history['ll_tilde']),
                      'log_lik_R': np.asarray(history['log_lik_R']),
                      'jump_chain_evolution': history['jump_chain_evolution'], 'best_epoch': best,
                      'best_log_lik': history['log_lik_R'][best] if len(elbos) else None,
                      'best_jump_chain': history['jump_chain_evolution'][best] if len(elbos) else None,
                      'best_newick': history['newick'][best] if len(elbos) else None}
        with open(os.path.join(save_dir, 'results.p'), 'wb') as f:
            pickle.dump(resultDict, f)
        return resultDict

    def train(self, epochs=100, batch_size=128, learning_rate=0.001, memory_optimization='on', save_dir='auto'):
        """vcsmc.py:466-645: per epoch, one optimiser step per site minibatch (all slices but the last,
        vcsmc.py:533), then the full-S evaluation sweep the reference reports (vcsmc.py:538-551); same printed
        lines and result artefacts (results/<dataset>/<nested>/<K>/<timestamp>/{run_parameters.txt,results.p};
        save_dir=None to skip; no plots).  Returns the per-epoch ELBOs."""
        self.lr = learning_rate
        data_view = np.broadcast_to(self.genome_NxSxA, (1,) + self.genome_NxSxA.shape)
        slices = self.batch_slices(data_view, batch_size)
        print('================= Dataset shape: KxNxSxA =================')
        print((self.K, self.N, self.S, self.A))
        print('==========================================================')
        ctx = self._context()                              # fixes the device (and joins the ranks when --n_gpus > 1)
        world, rank = getattr(self, '_world', 1), getattr(self, '_rank', 0)
        # --train_parallel replicas (the default with --n_gpus > 1): data-parallel training -- on every minibatch each rank sweeps
        # its OWN K-particle system(s) (own seeds) on its own GPU and the optimiser steps on the mean of all gradients, i.e.
        # world x grad_samples independent ELBO samples per step.  redundant: every rank takes the identical step.
        replicas = world > 1 and str(getattr(self.args, 'train_parallel', 'replicas') or 'replicas') == 'replicas'
        n_local = max(1, int(getattr(self.args, 'grad_samples', 1) or 1))   # particle systems per step and rank
        if world > 1:
            # all ranks must train on the SAME site minibatches; python's
            # global RNG (unseeded, like the reference) differs from process to process: rank 0's slices are everybody's
            flat = ctx.comm_allgather_blob(np.asarray(sum(slices, []), dtype=np.int32))[0]
            cuts = np.cumsum([len(sl) for sl in slices])[:-1]
            slices = [[int(v) for v in part] for part in np.split(flat, cuts)]
        self.optimizer = train_mod.make_optimizer(getattr(self.args, 'optimizer', ''), self.lr)   # vcsmc.py:488-491
        nested = bool(getattr(self.args, 'nested', False))
        trainer = None
        if len(slices) > 1:
            trainer = train_mod.Trainer(self.genome_NxSxA, self.K, self.variables, self.optimizer, len(slices[0]),
                                        device=self._device, nested=nested, M=self.M)
        initial = self.sample_phylogenies()
        print('===================\nInitial evaluation of ELBO:', round(initial, 3))
        print('Initial jump chain:')
        print(self.jump_chains[0])
        print('===================')
        print('Training begins --')
        elbos = []
        hist = {k: [] for k in ('cost', 'log_weights', 'Qmatrices', 'left_branches', 'right_branches', 'log_lik', 'll_tilde',
                                'log_lik_R', 'jump_chain_evolution', 'newick')}
        self.minibatch_costs = []
        try:
            for i in range(epochs):
                bt = datetime.now()
                if trainer is not None:
                    for j in range(len(slices) - 1):                       # vcsmc.py:533 (the last slice is never used)
                        seed = self.seed + self._sweeps
                        self._sweeps += 1
                        # sample i of the step (i = rank * grad_samples + g when the ranks train as replicas) draws from seed + i 2^32:
                        # seeds are 64-bit, so 2 ranks x 1 sample and 1 rank x 2 samples are the same training run bit for bit
                        first = rank * n_local if replicas else 0
                        seeds = [seed + ((first + g) << 32) for g in range(n_local)]
                        self.minibatch_costs.append(trainer.step(slices[j], seeds[0], seeds[1:], ctx if replicas else None))
                    self._sync_from_variables()
                elbo = self.sample_phylogenies()
                best_k = int(np.argmax(self.log_likelihood_R))
                for key, val in (('cost', elbo), ('log_weights', self.log_weights), ('Qmatrices', self.Qmatrix),
                                 ('left_branches', self.left_branches), ('right_branches', self.right_branches),
                                 ('log_lik', self.log_likelihood), ('ll_tilde', self.log_likelihood_tilde),
                                 ('log_lik_R', self.log_likelihood_R), ('jump_chain_evolution', self.jump_chains),
                                 ('newick', self.newick(best_k))):
                    hist[key].append(val)
                print('Epoch', i + 1)
                print('ELBO\n', round(elbo, 3))
                print('Stationary probabilities\n', self.stationary_probs)
                print('Q-matrix\n', self.Qmatrix)
                print('LB param:\n', self.left_branches_param)
                print('RB param:\n', self.right_branches_param)
                elbos.append(elbo)
                at = datetime.now()
                print('Time spent\n', at - bt, '\n-----------------------------------------')
        finally:
            if trainer is not None:
                trainer.close()
        print("Done training.")
        self.elbos = np.asarray(elbos)
        if save_dir is not None and getattr(self, '_rank', 0) == 0:   # sharded: every rank holds the same results; rank 0 writes
            if save_dir == 'auto':                   # vcsmc.py:504-507
                tm = str(datetime.now())
                save_dir = './results/' + str(getattr(self.args, 'dataset', 'data')) + '/' + str(getattr(self.args, 'nested', False)) + \
                    '/' + str(getattr(self.args, 'n_particles', self.K)) + '/' + (tm[:10] + '-' + tm[11:13] + tm[14:16] + tm[17:19]) + '/'
            self.results = self._save_results(save_dir, initial, hist)
            self.save_dir = save_dir
            print("Finished...")
        return self.elbos
