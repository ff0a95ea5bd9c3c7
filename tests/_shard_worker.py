"""One rank of a sharded sweep (helper process of tests/test_gpu_sharded.py and of the CPU protocol test).
usage: python tests/_shard_worker.py RANK WORLD K DATASET SEED JC OUT.npz [N_SWEEPS]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    rank, world, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dataset, seed, jc, out = sys.argv[4], int(sys.argv[5]), sys.argv[6] == '1', sys.argv[7]
    n_sweeps = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    from phylo_amd import _ffi, model
    from phylo_amd.datasets import load_dataset
    from phylo_amd.rendezvous import exchange_comm_id
    g = load_dataset(dataset)['genome']
    N, S, _ = g.shape
    Q = model.jc_Q() if jc else model.get_Q(model.init_y_q())
    pi = model.get_stationary_probs(np.zeros(4) + 0.25)
    lam = np.full(N - 1, 10.0)
    ctx = _ffi.Context(K, N, S, device=int(os.environ.get('PHYLO_TEST_DEVICE', '0')))
    ctx.set_leaves(g)
    ctx.set_model(Q, pi, lam, lam, jc69_closed_form=jc)
    cid = exchange_comm_id(rank, world, _ffi.comm_unique_id if rank == 0 else None)
    ctx.comm_init(rank, world, cid)
    res = None
    twist_M = int(os.environ.get('PHYLO_TEST_TWIST_M', '0'))
    flags = _ffi.FLAGS_DEFAULT | (_ffi.TWISTING if twist_M else 0)
    batch = int(os.environ.get('PHYLO_TEST_BATCH', '0'))
    if batch and int(os.environ.get('PHYLO_TEST_INFLIGHT', '1')) > 1:
        # bench.py's exact loop at N > 1: several contexts on ONE shared communicator, each carrying `batch` independent sweeps,
        # advanced rank event by rank event: first halves of all contexts (lazy nodes: marks, owners' writes, their barrier),
        # then the second halves (bookkeeping, merge, all-gather, scan); context i uses seeds seed + 100 i + 10 j
        inflight = int(os.environ['PHYLO_TEST_INFLIGHT'])
        group = [ctx]
        for i in range(1, inflight):
            c2 = _ffi.Context(K, N, S, device=int(os.environ.get('PHYLO_TEST_DEVICE', '0')))
            c2.set_leaves(g)
            c2.set_model(Q, pi, lam, lam, jc69_closed_form=jc)
            c2.comm_share(ctx)
            group.append(c2)
        for rep in range(n_sweeps):
            for i, c in enumerate(group):
                c.sweep_batch_begin([seed + 100 * i + 10 * j for j in range(batch)], flags=flags)
            for _ in range(N - 1):
                for c in group:
                    c.sweep_step_a()
                for c in group:
                    c.sweep_step()
            for c in group:
                c.sweep_finish()
        outs = {}
        for i, c in enumerate(group):
            r = c.sweep_fetch()
            outs['log_weights%d' % i] = r['log_weights']
            outs['ancestors%d' % i] = r['ancestors']
            outs['logz%d' % i] = c.sweep_fetch_logz(batch)
        np.savez(out, k0=ctx.k0, **outs)
        for c in reversed(group[1:]):
            c.close()
        ctx.close()
        return
    if batch:                                      # G independent sweeps in one sharded context (bench.py at N > 1)
        seeds = [seed + 10 * i for i in range(batch)]
        for rep in range(n_sweeps):
            ctx.sweep_batch_async(seeds, flags=flags)
        res = ctx.sweep_fetch()
        logz = ctx.sweep_fetch_logz(batch)
        np.savez(out, log_weights=res['log_weights'], log_likelihood=res['log_likelihood'], ancestors=res['ancestors'],
                 merges=res['merges'], logz=logz, k0=ctx.k0)
        ctx.close()
        return
    inflight = int(os.environ.get('PHYLO_TEST_INFLIGHT', '1'))
    others = []
    if inflight > 1:
        # several sweeps in flight on this rank: further contexts share ctx's communicator and all of them advance
        # rank event by rank event (bench.py's sharded loop).  Sweep i of the group uses seed + i; the LAST context's
        # result (seed + inflight - 1) is reported.
        for i in range(1, inflight):
            c2 = _ffi.Context(K, N, S, device=int(os.environ.get('PHYLO_TEST_DEVICE', '0')))
            c2.set_leaves(g)
            c2.set_model(Q, pi, lam, lam, jc69_closed_form=jc)
            c2.comm_share(ctx)
            others.append(c2)
        group = [ctx] + others
        for rep in range(n_sweeps):
            for i, c in enumerate(group):
                c.sweep_begin(seed + i, flags=flags, M=max(twist_M, 1))
            for _ in range(N - 1):
                if os.environ.get('PHYLO_TEST_GROUP_STEP'):
                    _ffi.sweep_step_group(group)      # one grouped collective per rank event (bench.py's loop)
                else:
                    for c in group:
                        c.sweep_step_a()              # first halves of all contexts, then the second halves
                    for c in group:
                        c.sweep_step()
            for c in group:
                c.sweep_finish()
        first = ctx.sweep_fetch()
        res = group[-1].sweep_fetch()
        res['first_logZ'] = first['logZ']
        ctx_report = group[-1]
    else:
        for s in range(n_sweeps):                     # back-to-back sweeps reuse the node pool slabs
            res = ctx.sweep(seed + s, flags=flags, M=max(twist_M, 1))
        ctx_report = ctx
    t = ctx.comm_max(float(rank))
    assert t == world - 1, t
    ctx.comm_barrier()
    cache_used, cache_cap = ctx_report.debug_remote_cache()    # (before sweep_node: it may run further kernels)
    node = ctx_report.sweep_node(N - 2, ctx.K_local - 1)
    np.savez(out, cache_used=cache_used, cache_cap=cache_cap, log_weights=res['log_weights'], log_likelihood=res['log_likelihood'], ancestors=res['ancestors'],
             merges=res['merges'], left_branches=res['left_branches'], logZ=res['logZ'], node=node, k0=ctx_report.k0,
             first_logZ=res.get('first_logZ', res['logZ']))
    for c in reversed(others):
        c.close()
    ctx.close()


if __name__ == '__main__':
    main()
