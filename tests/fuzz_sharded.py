"""Random sharded sweeps (2-4 processes on GPU 0, host-side collectives over shared memory) against the unsharded C oracle:
particles per rank, exchange form, lazy / eager nodes, the size of the local cache of remote nodes.
python tests/fuzz_sharded.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO                 # noqa: E402
from oracle import cpu_ref as O                   # noqa: E402
from phylo_amd.datasets import load_dataset       # noqa: E402
from tests.test_gpu_sharded import PI, run_world  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    world = int(rng.integers(2, 5))
    Kl = int(rng.choice([4, 16, 17, 64, 100, 256]))
    K = Kl * world
    dataset = str(rng.choice(['primate_data', 'primate_data_wang']))
    jc = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1000))
    env = {}
    if rng.integers(0, 3) == 0:
        env['PHYLO_EAGER_NODES'] = '1'
    if rng.integers(0, 3) == 0:
        env['PHYLO_P2P'] = '0'
    c = int(rng.integers(0, 4))
    if c == 0:
        env['PHYLO_NO_REMOTE_CACHE'] = '1'
    elif c == 1:
        env['PHYLO_REMOTE_CACHE_CAP'] = str(int(rng.integers(1, 6)))
    what = "world=%d K=%d %s jc=%s seed=%d env=%s" % (world, K, dataset, jc, seed, env)
    try:
        parts = run_world(world, K, dataset, seed, jc, n_sweeps=2, extra_env=env)
        g = load_dataset(dataset)['genome']
        N = g.shape[0]
        Q = O.jc_Q() if jc else O.get_Q(O.init_y_q())
        lam = np.full(N - 1, 10.0)
        ref = CO.sweep(g, Q, PI, lam, lam, K, seed + 1, jc=jc)
        for r, p in enumerate(parts):
            sl = slice(r * Kl, (r + 1) * Kl)
            np.testing.assert_array_equal(p['ancestors'], ref['ancestors'][:, sl])
            assert np.array_equal(p['log_weights'].view(np.uint64), ref['log_weights'][:, sl].view(np.uint64))
            assert float(p['logZ']) == ref['logZ']
    except Exception:
        print("FAILED:", what, flush=True)
        raise
    n += 1
    print("ok %d: %s (cache used %s of %s)" % (n, what, [int(p['cache_used']) for p in parts], int(parts[0]['cache_cap'])), flush=True)
print("fuzz_sharded: %d random sharded configurations bit-exact against the oracle" % n)
