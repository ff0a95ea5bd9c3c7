"""One rank of `runner.py --n_gpus WORLD` (helper of tests/test_gpu_sharded.py).
usage: python tests/_runner_worker.py RANK WORLD OUT.npz -- runner args..."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world))
    import random
    random.seed(3 if rank == 0 else 1000 + rank)     # batch_slices uses python's global RNG, like the reference: the
                                                     # ranks of a sharded run do not share its state, rank 0's slices must win
    import runner
    from phylo_amd.datasets import load_dataset
    from phylo_amd.vcsmc import VCSMC
    args = runner.parse_args(sys.argv[5:])
    v = VCSMC(load_dataset(args.dataset), K=args.n_particles, args=args)
    elbos = v.train(epochs=args.num_epoch, batch_size=args.batch_size, learning_rate=args.learning_rate,
                    save_dir=os.path.join(os.path.dirname(out), 'results') if rank == 0 else None)
    np.savez(out, elbos=elbos, lam=v.left_branches_param, newick=np.array(v.newick(0)), log_weights=v.log_weights,
             ancestors=v.ancestors)
    v.close()


if __name__ == '__main__':
    main()
