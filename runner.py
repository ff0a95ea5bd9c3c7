#!/usr/bin/env python3
"""runner.py -- the reference's CLI (runner.py:12-58, 61-212) over the MI355X library.

Same flags and defaults.  Differences: datasets come from an explicit table
(phylo_amd/datasets.py) instead of `exec(args.dataset + ' = True')`; `--twisting` is accepted as an alias of
`--nested` (the reference's README advertises it, its parser lacks it); `--seed`, `--n_gpus`, `--train_parallel`,
`--grad_samples` and `--ambiguity` (default: the reference's KeyError on characters such as DS7's 'N'; `iupac` encodes them) are new.
"""
import argparse

import numpy as np


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Variational Combinatorial Sequential Monte Carlo')
    parser.add_argument('--dataset', help='benchmark dataset to use.', default='primate_data')
    parser.add_argument('--n_particles', type=int, help='number of SMC samples.', default=10)
    parser.add_argument('--batch_size', type=int, help='number of sites on genome per batch.', default=256)
    parser.add_argument('--learning_rate', type=float, help='Learning rate.', default=0.001)
    parser.add_argument('--num_epoch', type=int, help='number of epoches to train.', default=100)
    parser.add_argument('--optimizer', type=str, help='Optimizer for Training', default='GradientDescentOptimizer')
    parser.add_argument('--branch_prior', type=float, help='Hyperparameter for branch length initialization.',
                        default=np.log(10))
    parser.add_argument('--M', type=int, help='number of subparticles to compute look-ahead particles', default=10)
    parser.add_argument('--nested', default=False, type=lambda x: (str(x).lower() == 'true'))
    parser.add_argument('--twisting', default=None, type=lambda x: (str(x).lower() == 'true'),
                        help='alias of --nested (README.md:28 of the reference)')
    parser.add_argument('--jcmodel', default=False, type=lambda x: (str(x).lower() == 'true'))
    parser.add_argument('--memory_optimization', help='Use memory optimization?', default='on')
    parser.add_argument('--seed', type=int, default=0, help='seed of the counter-based RNG contract')
    parser.add_argument('--n_gpus', type=int, default=1,
                        help='one process per GPU (python -m torch.distributed.run --nproc-per-node N runner.py --n_gpus N ...): the '
                             'particles of the EVALUATION sweeps are sharded over the ranks (global resampling, same bits as one GPU); '
                             'training: see --train_parallel')
    parser.add_argument('--train_parallel', choices=('replicas', 'redundant'), default='replicas',
                        help='with --n_gpus N > 1: replicas = data-parallel training, every rank sweeps its own n_particles-particle '
                             'system per minibatch (own seed) and the optimiser steps on the mean gradient of the N ranks; '
                             'redundant = every rank takes the identical step (equals the one-process run bit for bit)')
    parser.add_argument('--grad_samples', type=int, default=1,
                        help='independent particle systems swept per optimiser step (and per rank with --train_parallel replicas); the '
                             'step is taken on their mean gradient')
    parser.add_argument('--ambiguity', choices=('error', 'iupac'), default='error',
                        help="characters outside the dataset's alphabet: KeyError like the reference, or IUPAC indicator rows")
    args = parser.parse_args(argv)
    if args.twisting is not None:
        args.nested = args.twisting
    return args


def main(argv=None):
    args = parse_args(argv)
    from phylo_amd.datasets import load_dataset
    from phylo_amd.vcsmc import VCSMC
    datadict = load_dataset(args.dataset, ambiguity=args.ambiguity)
    vcsmc = VCSMC(datadict, K=args.n_particles, args=args)
    return vcsmc.train(epochs=args.num_epoch, batch_size=args.batch_size, learning_rate=args.learning_rate,
                       memory_optimization=args.memory_optimization)


if __name__ == "__main__":
    main()
