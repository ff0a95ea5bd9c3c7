"""The N > 1 path on CPU: world_size-2 processes over torch.distributed (gloo) run the sharded protocol
(oracle/sharded_ref.py: root tables replicated or kept by their owner, one all-gather of three K-vectors per rank event, remote
nodes fetched from their owner only when merged) and must reproduce the unsharded oracle sweep; also the
file rendezvous that hands the RCCL id to the ranks."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, K, seed, tmp, local_tables, lazy, flags=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import cpu_ref as O
    from oracle.sharded_ref import sweep_sharded
    from phylo_amd.datasets import load_dataset
    from phylo_amd.rendezvous import exchange_comm_id

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['PHYLO_RDZV_DIR'] = tmp
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # the id hand-off used by bench.py / the shard workers
    cid = exchange_comm_id(rank, world, (lambda: bytes(range(128))) if rank == 0 else None)
    assert cid[:128] == bytes(range(128))

    class Comm:
        """all_gather over gloo; remote nodes: every rank publishes the nodes it owns that somebody may
        read (test simplicity: an all_gather of this rank's node dict keys on demand would need an RPC; the
        GPU path reads the owner's pool in place instead)."""

        def __init__(self):
            self.pool = None

        def all_gather(self, arr):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return [o.numpy() for o in out]

        def gather_tables(self, mine):
            objs = [None] * world
            dist.all_gather_object(objs, mine)
            return objs

        def barrier(self):
            dist.barrier()

        def serve_begin(self, pool):
            # exchange every rank's full pool (small test sizes) so that fetch_node can be answered locally
            objs = [None] * world
            dist.all_gather_object(objs, pool)
            self.pools = objs

        def serve_end(self):
            self.pools = None

        def fetch_node(self, owner, key):
            return self.pools[owner][key]

    class FlagComm(Comm):
        """The device-side exchange of the GPU path (pk_p2p_exchange) as a protocol model: every rank owns a slab (a file
        mapped by all ranks, like the hipIpc mappings) holding R rows of 3 x K doubles and two sets of flags; an exchange writes
        this rank's segment of row r into EVERY peer's slab, then its flag (the monotone epoch of the exchange) in every peer's
        slab, and waits until every peer's flag in the OWN slab has reached the epoch.  Rows are reused from sweep to sweep
        (one epoch counter per purpose, as on the GPU); the barrier is the same exchange without a payload."""

        def __init__(self, R):
            super().__init__()
            self.R, self.row, self.epoch = R, 0, [0, 0]
            self.words = R * 3 * K + 2 * world
            mine = np.memmap(os.path.join(tmp, 'slab%d.bin' % rank), dtype=np.float64, mode='w+', shape=(self.words,))
            mine[:] = 0.0
            mine.flush()
            dist.barrier()
            self.slabs = [np.memmap(os.path.join(tmp, 'slab%d.bin' % p), dtype=np.float64, mode='r+', shape=(self.words,))
                          for p in range(world)]

        def _flags(self, p, purpose):
            return self.slabs[p][self.R * 3 * K + purpose * world:self.R * 3 * K + (purpose + 1) * world]

        def _signal_and_wait(self, purpose):
            import time
            self.epoch[purpose] += 1
            e = float(self.epoch[purpose])
            for p in range(world):
                if p != rank:
                    self._flags(p, purpose)[rank] = e
            t0 = time.time()
            while any(self._flags(rank, purpose)[p] < e for p in range(world) if p != rank):
                assert time.time() - t0 < 60, "flag wait timed out"
                time.sleep(0)

        def all_gather(self, arr):
            r, Kl = self.row % self.R, arr.shape[1]
            self.row += 1
            for p in range(world):                           # my segment of the three vectors into every slab (my own included)
                v = self.slabs[p][r * 3 * K:(r + 1) * 3 * K].reshape(3, K)
                v[:, rank * Kl:(rank + 1) * Kl] = arr
            self._signal_and_wait(0)
            full = np.array(self.slabs[rank][r * 3 * K:(r + 1) * 3 * K].reshape(3, K))
            return [full[:, p * Kl:(p + 1) * Kl] for p in range(world)]

        def barrier(self):
            self._signal_and_wait(1)

    g = load_dataset('primate_data_wang')['genome'][:, :120]
    N = g.shape[0]
    Q, pi, lam = O.get_Q(O.init_y_q()), np.full((1, 4), 0.25), np.full(N - 1, 10.0)
    comm = FlagComm(N - 1) if flags else Comm()
    if flags:                                                # a first sweep with another seed: the rows and flags are reused
        sweep_sharded(comm, rank, world, g, Q, pi, lam, lam, K, seed + 1, local_tables=local_tables, lazy=lazy)
    out = sweep_sharded(comm, rank, world, g, Q, pi, lam, lam, K, seed, local_tables=local_tables, lazy=lazy)
    ref = O.sweep(g, Q, pi, lam, lam, K, seed)
    np.testing.assert_array_equal(out['ancestors'], ref['ancestors'])
    np.testing.assert_allclose(out['log_weights'], ref['log_weights'], rtol=1e-12)
    assert abs(out['logZ'] - ref['logZ']) < 1e-9 * abs(ref['logZ'])
    # the local cache of remote nodes: unbounded, every remote node crosses once; two slots: same results, the overflow is read
    # in place at every merge; none (the form before the cache): more fetches than nodes as soon as a node is merged twice
    assert out['remote_fetches'] == out['cached_nodes'] and out['cache_overflow'] == 0
    if not flags:
        tiny = sweep_sharded(comm, rank, world, g, Q, pi, lam, lam, K, seed, local_tables=local_tables, lazy=lazy, cache_slots=2)
        none = sweep_sharded(comm, rank, world, g, Q, pi, lam, lam, K, seed, local_tables=local_tables, lazy=lazy, cache_slots=0)
        for o in (tiny, none):
            np.testing.assert_array_equal(o['ancestors'], ref['ancestors'])
            assert np.array_equal(o['log_weights'], out['log_weights']) and o['logZ'] == out['logZ']
        assert tiny['cached_nodes'] <= 2 and none['cached_nodes'] == 0
        assert none['remote_fetches'] >= tiny['remote_fetches'] >= out['remote_fetches']
    np.save(os.path.join(tmp, 'fetch%d.npy' % rank), np.array([out['remote_fetches']]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,seed,local_tables,lazy,flags", [(16, 0, False, False, False), (24, 3, True, False, False),
                                                            (24, 5, True, True, False), (24, 5, True, 'draws', False),
                                                            (40, 9, True, 'draws', False), (24, 5, True, 'draws', True),
                                                            (16, 2, False, False, True)])
def test_sharded_protocol_world2(K, seed, local_tables, lazy, flags):
    """flags=True: the all-gather and the barrier go through the slab-and-flag exchange (the GPU path's pk_p2p_exchange) over
    files mapped by both processes; two sweeps in a row reuse rows and flags."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() * 7 + K + 13 * int(bool(flags))) % 1000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, K, seed, tmp, local_tables, lazy, flags), nprocs=2, join=True)
        fetched = sum(int(np.load(os.path.join(tmp, 'fetch%d.npy' % r))[0]) for r in range(2))
        assert fetched > 0, "the test never exercised a remote child"


def _train_worker(rank, world, port, tmp, steps):
    """Data-parallel training on CPU: the gradient of each rank's own particle system from the gradient oracle, the product's
    mean over ranks (phylo_amd/train.py: mean_of_samples) with the all-gather over gloo, the product's Adam."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import cpu_grad as G
    from phylo_amd import train as T
    from phylo_amd.datasets import load_dataset

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)

    class Ctx:
        def comm_allgather_blob(self, a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return np.stack([o.numpy() for o in out])

    g = load_dataset('primate_data_wang')['genome'][:5, :40]
    N, K = g.shape[0], 8
    v = T.Variables(N, np.log(10.0), False)
    opt = T.make_optimizer('Adam', 0.05)
    for step in range(steps):
        Q, pi, ll, lr = v.evaluate()
        raw = G.sweep_grad(g, Q, pi, ll, lr, K, 100 + step + (rank << 32), 0)
        grads = T.chain_rules(v, Q, pi, ll, lr, raw)
        logZ, grads = T.mean_of_samples([(raw['logZ'], grads)], Ctx())
        opt.apply(v, grads)
    np.savez(os.path.join(tmp, 'v%d.npz' % rank), **{n: getattr(v, n) for n in v.names()}, logZ=logZ)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_world2_equals_one_process_with_two_samples():
    """Two ranks, one particle system each per step == one process, two systems per step: same variables bit for bit after three
    Adam steps, and both ranks hold the same bits."""
    import torch.multiprocessing as mp
    from oracle import cpu_grad as G
    from phylo_amd import train as T
    from phylo_amd.datasets import load_dataset
    steps = 3
    port = 29500 + (os.getpid() * 11 + 77) % 1000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_train_worker, args=(2, port, tmp, steps), nprocs=2, join=True)
        got = [dict(np.load(os.path.join(tmp, 'v%d.npz' % r))) for r in range(2)]
    g = load_dataset('primate_data_wang')['genome'][:5, :40]
    N, K = g.shape[0], 8
    v = T.Variables(N, np.log(10.0), False)
    opt = T.make_optimizer('Adam', 0.05)
    for step in range(steps):
        Q, pi, ll, lr = v.evaluate()
        samples = []
        for i in range(2):
            raw = G.sweep_grad(g, Q, pi, ll, lr, K, 100 + step + (i << 32), 0)
            samples.append((raw['logZ'], T.chain_rules(v, Q, pi, ll, lr, raw)))
        logZ, grads = T.mean_of_samples(samples)
        opt.apply(v, grads)
    for n in v.names():
        for r in range(2):
            assert np.array_equal(np.asarray(getattr(v, n)).view(np.uint64), got[r][n].view(np.uint64)), (n, r)
    assert float(got[0]['logZ']) == float(got[1]['logZ']) == logZ
