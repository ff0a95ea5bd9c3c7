"""Sharded sweep on real hardware.  A one-GPU box cannot host several RCCL ranks (RCCL refuses duplicate
devices), so several PROCESSES share GPU 0; PHYLO_COMM=hostshm carries the host-side collectives (rendezvous of the hipIpc
handles, phylo_comm_max).  The exchange of every rank event is the product's own device-side one (pk_p2p_exchange: writes
into the peers' hipIpc-mapped slabs + flags; the default), or with PHYLO_P2P=0 the collective path (hostshm here, RCCL in
production); the sharded bookkeeping, the node addressing and the hipIpc peer mappings of the node pools are the product's own.  The result must be bit-identical to the unsharded sweep and to the oracle.
RCCL itself is exercised with a world of one rank."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cpu_ref as O
from phylo_amd import _ffi
from phylo_amd.datasets import load_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PI = np.full((1, 4), 0.25)


def run_world(world, K, dataset, seed, jc, n_sweeps=1, transport='hostshm', extra_env=None):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PHYLO_RDZV_DIR=tmp, MASTER_PORT=str(29000 + os.getpid() % 2000), PHYLO_COMM=transport)
        env.update(extra_env or {})
        procs = []
        for r in range(world):
            out = os.path.join(tmp, "r%d.npz" % r)
            procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), str(r),
                                                 str(world), str(K), dataset, str(seed), '1' if jc else '0', out,
                                                 str(n_sweeps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        outs = []
        for out, p in procs:
            try:
                log, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for _, q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, log.decode()[-2000:]
            outs.append(dict(np.load(out)))
        return outs


@pytest.mark.parametrize("world,K,jc,form", [(2, 64, True, 'default'), (3, 96, False, 'default'), (2, 64, False, 'replicated'),
                                             (2, 64, False, 'eager'), (3, 96, True, 'eager'),
                                             (2, 64, True, 'collective'), (3, 96, False, 'collective'), (2, 64, False, 'eager-collective'),
                                             (2, 64, False, 'no-cache'), (3, 96, False, 'tiny-cache'), (2, 64, False, 'eager-tiny-cache')])
def test_sharded_sweep_bit_identical(world, K, jc, form):
    """default: lazy nodes, every rank advances only its own particles' root tables and reads an adopted ancestor's rows
    from the owner's slab (peer mapping), adopted nodes are marked from the replicated index search and written by their
    owner before a barrier collective; eager (PHYLO_EAGER_NODES=1): every node stored, one collective per rank event;
    replicated (PHYLO_REPLICATED_BOOK=1): every rank advances all K tables redundantly."""
    dataset, seed, n_sweeps = 'primate_data', 4, 2
    env = {'replicated': {'PHYLO_REPLICATED_BOOK': '1'}, 'eager': {'PHYLO_EAGER_NODES': '1'},
           'collective': {'PHYLO_P2P': '0'},                # the all-gather per rank event instead of the device-side exchange
           'eager-collective': {'PHYLO_EAGER_NODES': '1', 'PHYLO_P2P': '0'},
           'no-cache': {'PHYLO_NO_REMOTE_CACHE': '1'},      # every remote child read in place (the form before the cache)
           'tiny-cache': {'PHYLO_REMOTE_CACHE_CAP': '3'},   # the cache fills up: the rest is read in place
           'eager-tiny-cache': {'PHYLO_EAGER_NODES': '1', 'PHYLO_REMOTE_CACHE_CAP': '2'}}.get(form)
    parts = run_world(world, K, dataset, seed, jc, n_sweeps=n_sweeps, extra_env=env)
    g = load_dataset(dataset)['genome']
    N = g.shape[0]
    Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ref = CO.sweep(g, Q, PI, lam, lam, K, seed + n_sweeps - 1, jc=jc, want_nodes=True)
    Kl = K // world
    for r, p in enumerate(parts):
        assert int(p['k0']) == r * Kl
        sl = slice(r * Kl, (r + 1) * Kl)
        np.testing.assert_array_equal(p['ancestors'], ref['ancestors'][:, sl])          # global indices, bit-exact
        np.testing.assert_array_equal(p['merges'], ref['merges'][:, sl])
        assert np.array_equal(p['log_weights'].view(np.uint64), ref['log_weights'][:, sl].view(np.uint64))
        assert np.array_equal(p['log_likelihood'].view(np.uint64), ref['log_likelihood'][:, sl].view(np.uint64))
        assert float(p['logZ']) == ref['logZ']                                           # every rank holds the global log Z
        assert np.array_equal(p['node'].view(np.uint64), ref['nodes'][N - 2, (r + 1) * Kl - 1].view(np.uint64))
    # remote children were really exercised: some ancestor of a rank-0 particle lives on another rank
    assert (parts[0]['ancestors'] >= Kl).any()
    # ... and went through the local cache of remote nodes (fetched once per sweep), unless it is switched off / too small
    used, cap = [int(p['cache_used']) for p in parts], [int(p['cache_cap']) for p in parts]
    if form == 'no-cache':
        assert cap == [0] * world and used == [0] * world
    elif form == 'replicated':                               # (bookkeeping ahead of the owners' writes: remote nodes in place)
        assert used == [0] * world
    elif 'tiny' in form:
        assert max(used) > cap[0] > 0, (used, cap)           # more nodes wanted than slots: the overflow path ran
    else:
        assert min(cap) > 0 and max(used) > 0, (used, cap)


def test_sharded_lazy_nodes_bit_identical():
    """Lazy nodes when sharded: the owner writes an adopted node, a barrier collective orders it before the
    peers' merges; phylo_sweep_node completes the pool collectively."""
    world, K, seed = 2, 64, 7
    parts = run_world(world, K, 'primate_data', seed, False, n_sweeps=2, extra_env={'PHYLO_LAZY_NODES': '1'})
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ref = CO.sweep(g, Q, PI, lam, lam, K, seed + 1, want_nodes=True)
    Kl = K // world
    for r, p in enumerate(parts):
        sl = slice(r * Kl, (r + 1) * Kl)
        np.testing.assert_array_equal(p['ancestors'], ref['ancestors'][:, sl])
        assert np.array_equal(p['log_weights'].view(np.uint64), ref['log_weights'][:, sl].view(np.uint64))
        assert float(p['logZ']) == ref['logZ']
        assert np.array_equal(p['node'].view(np.uint64), ref['nodes'][N - 2, (r + 1) * Kl - 1].view(np.uint64))


@pytest.mark.parametrize("p2p", ['1', '0'])
def test_sharded_twisted_sweep_bit_identical(p2p):
    """Twisting when sharded: potentials are computed for local particles only, the chosen (pair, sub-sample)
    of every particle travels with one more exchange, then every rank updates all root tables."""
    world, K, M, seed = 2, 32, 2, 3
    parts = run_world(world, K, 'primate_data_wang', seed, True, extra_env={'PHYLO_TEST_TWIST_M': str(M), 'PHYLO_P2P': p2p})
    g = load_dataset('primate_data_wang')['genome']
    lam = np.full(8, 10.0)
    ref = CO.sweep_twisted(g, O.jc_Q(), PI, lam, lam, K, M, seed, jc=True)
    Kl = K // world
    for r, p in enumerate(parts):
        sl = slice(r * Kl, (r + 1) * Kl)
        np.testing.assert_array_equal(p['ancestors'], ref['ancestors'][:, sl])
        np.testing.assert_array_equal(p['merges'], ref['merges'][:, sl])
        assert np.array_equal(p['log_weights'].view(np.uint64), ref['log_weights'][:, sl].view(np.uint64))
        assert float(p['logZ']) == ref['logZ']


@pytest.mark.parametrize("rehearse", [False, True])
def test_rccl_single_rank_world(rehearse):
    """RCCL communicator with one rank: ncclCommInitRank / grouped in-place all-gather on the real library.  With
    PHYLO_REHEARSE_SHARDED=1 the one rank runs the complete sharded protocol (owner-held tables through the slab pointers,
    the index search of all particles that marks adopted nodes, the barrier collective before the merges)."""
    env = {'PHYLO_COMM_FORCE_RCCL': '1'}
    if rehearse:
        env['PHYLO_REHEARSE_SHARDED'] = '1'
    parts = run_world(1, 32, 'primate_data_wang', 1, True, n_sweeps=2, transport='rccl', extra_env=env)
    g = load_dataset('primate_data_wang')['genome']
    lam = np.full(8, 10.0)
    ref = CO.sweep(g, O.jc_Q(), PI, lam, lam, 32, 2, jc=True, want_nodes=True)
    assert np.array_equal(parts[0]['log_weights'].view(np.uint64), ref['log_weights'].view(np.uint64))
    np.testing.assert_array_equal(parts[0]['ancestors'], ref['ancestors'])
    assert float(parts[0]['logZ']) == ref['logZ']
    assert np.array_equal(parts[0]['node'].view(np.uint64), ref['nodes'][7, 31].view(np.uint64))


@pytest.mark.parametrize("transport,world,extra", [
    ('hostshm', 2, {}),                                   # two processes on GPU 0, three sweeps in flight each
    ('hostshm', 2, {'PHYLO_TEST_GROUP_STEP': '1'}),
    ('rccl', 1, {'PHYLO_COMM_FORCE_RCCL': '1'}),          # the real library on its dedicated comm stream
    ('rccl', 1, {'PHYLO_COMM_FORCE_RCCL': '1', 'PHYLO_TEST_GROUP_STEP': '1'}),   # ... one grouped all-gather per rank event
])
def test_sweeps_in_flight_share_one_communicator(transport, world, extra):
    """bench.py's sharded loop: contexts joined by phylo_comm_share advance rank event by rank event; every sweep
    equals the oracle's sweep of its seed."""
    K, seed, inflight = 48, 11, 3
    env = dict(extra, PHYLO_TEST_INFLIGHT=str(inflight))
    parts = run_world(world, K, 'primate_data', seed, False, n_sweeps=2, transport=transport, extra_env=env)
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    ref_last = CO.sweep(g, Q, PI, lam, lam, K, seed + inflight - 1, want_nodes=True)
    ref_first = CO.sweep(g, Q, PI, lam, lam, K, seed)
    Kl = K // world
    for r, p in enumerate(parts):
        sl = slice(r * Kl, (r + 1) * Kl)
        np.testing.assert_array_equal(p['ancestors'], ref_last['ancestors'][:, sl])
        assert np.array_equal(p['log_weights'].view(np.uint64), ref_last['log_weights'][:, sl].view(np.uint64))
        assert float(p['logZ']) == ref_last['logZ']
        assert float(p['first_logZ']) == ref_first['logZ']
        assert np.array_equal(p['node'].view(np.uint64), ref_last['nodes'][N - 2, (r + 1) * Kl - 1].view(np.uint64))


@pytest.mark.parametrize("world,G,Kg,copy_words", [(2, 3, 32, None), (3, 2, 48, None), (3, 2, 48, '64')])
def test_batched_sweeps_on_sharded_contexts(world, G, Kg, copy_words):
    """G independent sweeps in ONE sharded context: the G * Kg particle indices are sharded by contiguous ranges (a
    group straddles ranks when world does not divide G), one all-gather per rank event carries all of them, and
    every group is bit for bit the Kg-particle sweep of its seed."""
    seed = 6
    env = {'PHYLO_TEST_BATCH': str(G)}
    if copy_words:                                         # the large-exchange form: copy over many workgroups, then the flags alone
        env['PHYLO_P2P_COPY_WORDS'] = copy_words
    parts = run_world(world, G * Kg, 'primate_data', seed, False, n_sweeps=2, extra_env=env)
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    refs = [CO.sweep(g, Q, PI, lam, lam, Kg, seed + 10 * i) for i in range(G)]
    flat = {k: np.concatenate([r[k] for r in refs], axis=1) for k in ('log_weights', 'log_likelihood', 'ancestors', 'merges')}
    Kl = G * Kg // world
    for r, p in enumerate(parts):
        sl = slice(r * Kl, (r + 1) * Kl)
        np.testing.assert_array_equal(p['ancestors'], flat['ancestors'][:, sl])       # indices inside the group
        np.testing.assert_array_equal(p['merges'], flat['merges'][:, sl])
        assert np.array_equal(p['log_weights'].view(np.uint64), flat['log_weights'][:, sl].view(np.uint64))
        assert np.array_equal(p['log_likelihood'].view(np.uint64), flat['log_likelihood'][:, sl].view(np.uint64))
        assert list(p['logz']) == [ref['logZ'] for ref in refs]                        # every rank holds every estimate


@pytest.mark.parametrize("transport,world,G,Kg", [('hostshm', 2, 3, 32), ('rccl', 1, 3, 32),
                                                  ('hostshm', 2, 2, 8192), ('rccl', 1, 2, 8192)])
def test_bench_loop_at_n_gt_1(transport, world, G, Kg):
    """bench.py's exact N > 1 loop: 2 ranks x 2 contexts on one shared communicator x 3 batched sweeps per context,
    phylo_sweep_batch_begin + phylo_sweep_step_a + phylo_sweep_step + phylo_sweep_finish, lazy nodes (the default): every one
    of the 6 sweeps equals the oracle's sweep of its seed.  Also with the real library on a one-rank world running the complete
    sharded protocol.  Kg = 8192: the replicated scan of large groups, and the owners find their adopted nodes with one workgroup
    per 64 particles (pk_materialize_by_draws, grouped form), as bench.py's launch sets of 20 480 particles per rank do."""
    seed, inflight = 21, 2
    env = {'PHYLO_TEST_BATCH': str(G), 'PHYLO_TEST_INFLIGHT': str(inflight)}
    if transport == 'rccl':
        env.update(PHYLO_COMM_FORCE_RCCL='1', PHYLO_REHEARSE_SHARDED='1')
    parts = run_world(world, G * Kg, 'primate_data', seed, False, n_sweeps=2, transport=transport, extra_env=env)
    g = load_dataset('primate_data')['genome']
    N = g.shape[0]
    Q = O.get_Q(O.init_y_q())
    lam = np.full(N - 1, 10.0)
    Kl = G * Kg // world
    for i in range(inflight):
        refs = [CO.sweep(g, Q, PI, lam, lam, Kg, seed + 100 * i + 10 * j) for j in range(G)]
        lw = np.concatenate([r['log_weights'] for r in refs], axis=1)
        anc = np.concatenate([r['ancestors'] for r in refs], axis=1)
        for r, p in enumerate(parts):
            sl = slice(r * Kl, (r + 1) * Kl)
            assert np.array_equal(p['log_weights%d' % i].view(np.uint64), lw[:, sl].view(np.uint64)), (i, r)
            np.testing.assert_array_equal(p['ancestors%d' % i], anc[:, sl])
            assert list(p['logz%d' % i]) == [ref['logZ'] for ref in refs]


def test_stepwise_sweep_equals_whole_sweep():
    """phylo_sweep_begin / step / finish, interleaved over two unsharded contexts, against phylo_sweep."""
    g = load_dataset('primate_data_wang')['genome']
    N = g.shape[0]
    lam = np.full(N - 1, 10.0)
    cs = []
    for _ in range(2):
        c = _ffi.Context(40, N, g.shape[1])
        c.set_leaves(g)
        c.set_model(O.jc_Q(), PI, lam, lam, jc69_closed_form=True)
        cs.append(c)
    whole = [c.sweep(5 + i) for i, c in enumerate(cs)]
    for i, c in enumerate(cs):
        c.sweep_begin(5 + i)
    with pytest.raises(_ffi.PhyloError):
        cs[0].sweep_finish()                               # not all rank events issued yet
    for _ in range(N - 1):
        for c in cs:
            c.sweep_step()
    with pytest.raises(_ffi.PhyloError):
        cs[0].sweep_step()                                 # one too many
    for c in cs:
        c.sweep_finish()
    for i, c in enumerate(cs):
        out = c.sweep_fetch()
        assert out['logZ'] == whole[i]['logZ']
        assert np.array_equal(out['log_weights'].view(np.uint64), whole[i]['log_weights'].view(np.uint64))
        np.testing.assert_array_equal(out['ancestors'], whole[i]['ancestors'])
        c.close()


def test_comm_init_rejects_bad_world():
    g = load_dataset('primate_data_wang')['genome']
    ctx = _ffi.Context(10, 9, 738)
    with pytest.raises(_ffi.PhyloError):
        ctx.comm_init(0, 3, b'\0' * 128)          # K not divisible by world
    ctx.close()


def _run_runner(world, argv, port_salt):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PHYLO_RDZV_DIR=tmp, MASTER_PORT=str(29100 + os.getpid() % 800 + port_salt), PHYLO_COMM='hostshm')
        procs = []
        for r in range(world):
            out = os.path.join(tmp, "w%d.npz" % r)
            cmd = [sys.executable, os.path.join(ROOT, "tests", "_runner_worker.py"), str(r), str(world), out, '--'] + argv + \
                  ['--n_gpus', str(world)]
            procs.append((out, subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        res = []
        for out, p in procs:
            log, _ = p.communicate(timeout=240)
            assert p.returncode == 0, log.decode()[-2000:]
            res.append(dict(np.load(out)))
        return res


def test_runner_replicas_train_data_parallel():
    """`runner.py --n_gpus 2` in its default training mode (--train_parallel replicas; two processes on GPU 0, hostshm): every
    rank sweeps its own particle system per minibatch, the optimiser steps on the mean gradient -- both ranks end with the same
    bits, and they are the bits of ONE process that takes both samples itself (--grad_samples 2); the steps differ from a
    one-sample run's."""
    argv = ['--dataset', 'primate_data_wang', '--n_particles', '32', '--num_epoch', '2', '--batch_size', '256',
            '--optimizer', 'Adam', '--learning_rate', '0.05', '--seed', '4']
    two = _run_runner(2, argv, 11)
    one2 = _run_runner(1, argv + ['--grad_samples', '2'], 12)[0]
    one1 = _run_runner(1, argv, 13)[0]
    for r in two:
        assert np.array_equal(r['elbos'], one2['elbos'])
        assert np.array_equal(r['lam'].view(np.uint64), one2['lam'].view(np.uint64))
        assert np.array_equal(r['log_weights'].view(np.uint64), one2['log_weights'].view(np.uint64))
        np.testing.assert_array_equal(r['ancestors'], one2['ancestors'])
    assert not np.array_equal(one1['lam'], one2['lam'])
    # 2 ranks x 2 samples = 1 rank x 4 samples
    four = _run_runner(2, argv + ['--grad_samples', '2'], 14)
    one4 = _run_runner(1, argv + ['--grad_samples', '4'], 15)[0]
    for r in four:
        assert np.array_equal(r['lam'].view(np.uint64), one4['lam'].view(np.uint64))
        assert np.array_equal(r['elbos'], one4['elbos'])


def test_runner_n_gpus_equals_single_process():
    """`runner.py --n_gpus 2 --train_parallel redundant` (two processes on GPU 0, hostshm): the particles are sharded for the
    evaluation sweeps, every rank takes the same optimiser steps, and ELBOs, parameters and trees equal the one-process run."""
    argv = ['--dataset', 'primate_data_wang', '--n_particles', '32', '--num_epoch', '2', '--batch_size', '256',
            '--optimizer', 'Adam', '--learning_rate', '0.05', '--seed', '4', '--train_parallel', 'redundant']
    outs = {}
    for world in (1, 2):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, PHYLO_RDZV_DIR=tmp, MASTER_PORT=str(29100 + os.getpid() % 800 + world), PHYLO_COMM='hostshm')
            procs = []
            for r in range(world):
                out = os.path.join(tmp, "w%d.npz" % r)
                cmd = [sys.executable, os.path.join(ROOT, "tests", "_runner_worker.py"), str(r), str(world), out, '--'] + argv + \
                      ['--n_gpus', str(world)]
                procs.append((out, subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            res = []
            for out, p in procs:
                log, _ = p.communicate(timeout=240)
                assert p.returncode == 0, log.decode()[-2000:]
                res.append(dict(np.load(out)))
            if world == 2:
                assert os.path.exists(os.path.join(tmp, 'results', 'results.p'))
            outs[world] = res
    one, two = outs[1][0], outs[2]
    for r in two:
        assert np.array_equal(r['elbos'], one['elbos'])
        assert np.array_equal(r['lam'], one['lam'])
        assert str(r['newick']) == str(one['newick'])
        assert np.array_equal(r['log_weights'].view(np.uint64), one['log_weights'].view(np.uint64))
        np.testing.assert_array_equal(r['ancestors'], one['ancestors'])


def _run_bench_ranks(world, extra_env, args):
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for r in range(world):
            env = dict(os.environ, PHYLO_RDZV_DIR=tmp, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29300 + os.getpid() % 600), PHYLO_COMM='hostshm',
                       RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world))
            env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), '--gpus', str(world)] + args, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        outs = []
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err.decode()[-2000:]
            outs.append(out.decode())
        lines = [ln for ln in outs[0].splitlines() if ln.startswith('{')]
        assert len(lines) == 1 and not any(ln.startswith('{') for o in outs[1:] for ln in o.splitlines())   # rank 0 prints the ONE line
        import json
        return json.loads(lines[0])


def test_bench_py_at_two_ranks_reports_the_sharded_value_and_the_independent_figure():
    """bench.py itself as the driver launches it at N = 2 (two processes on GPU 0, host-side collectives over hostshm): the
    value is the sharded sweep's, 1 == N parity against the oracle holds, the independent-sweeps figure stands beside it."""
    line = _run_bench_ranks(2, {}, ['--steps', '12', '--warmup', '2', '--n_particles', '64', '--min-timed-ms', '20'])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    mg = line['multi_gpu']
    assert mg['form'].startswith("one sweep's particles sharded") and mg['exchange'] == 'p2p' and mg['fallback'] is None
    assert mg['independent_sweeps']['value'] > 0
    assert line['delta_logZ_max'] == 0.0 and line['ancestors_equal'] is True
    assert 'K_total=128' in line['config']['workload']


@pytest.mark.parametrize("fail,form,exchange", [('default', "one sweep's particles sharded", 'hostshm'),
                                                ('default,collective', 'independent sweeps per rank', 'none')])
def test_bench_py_falls_back_form_by_form(fail, form, exchange):
    """First contact gone wrong (forced): the device-side exchange fails -> every rank rebuilds on the collective path; that fails too
    -> independent sweeps per rank, the ranks meeting through files; the line says what happened and still carries a value."""
    line = _run_bench_ranks(2, {'PHYLO_BENCH_FAIL': fail}, ['--steps', '12', '--warmup', '2', '--n_particles', '64', '--min-timed-ms', '20'])
    mg = line['multi_gpu']
    assert mg['form'].startswith(form) and mg['exchange'] == exchange, mg
    assert 'forced failure' in mg['fallback'] and line['value'] > 0
    if form.startswith('independent'):
        assert 'independent sweeps of K=64' in line['config']['parallelism'] and 'K_total=64' in line['config']['workload']
        assert line['delta_logZ_max'] == 0.0
