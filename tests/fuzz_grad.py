"""Random small configurations of the reverse pass (plain and twisted proposal), GPU against oracle/cpu_grad.py at relative 1e-9.
python tests/fuzz_grad.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_grad as T     # noqa: E402  (its _check / _check_twisted do the comparison)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t0, n, kinds, last = time.time(), 0, {}, time.time()
while time.time() - t0 < budget:
    N = int(rng.integers(2, 9))
    S = int(rng.choice([1, 5, 64, 65, 130, 257, 300]))
    twisted = bool(rng.integers(0, 2))
    K = int(rng.integers(1, 25 if twisted else 60))
    M = int(rng.integers(1, 5)) if twisted else 1
    generic = rng.integers(0, 5) == 0
    genome = rng.uniform(0.05, 1.0, size=(N, S, 4)) if generic else T._codes_genome(rng, N, S)
    Q, pi, ll, lr = T._model(rng, N, spread=float(rng.uniform(0.1, 0.5)), lam=float(rng.uniform(1.0, 2.5)))
    seed = int(rng.integers(0, 2 ** 31))
    what = "N=%d S=%d K=%d %s M=%d generic=%s seed=%d" % (N, S, K, 'twisted' if twisted else 'plain', M, generic, seed)
    try:
        if twisted:
            T._check_twisted(genome, Q, pi, ll, lr, K=K, M=M, seed=seed)
        else:
            T._check(genome, Q, pi, ll, lr, K=K, seed=seed, flags=int(rng.integers(0, 2)))
    except AssertionError as e:
        print("FAILED:", what, str(e)[:300], flush=True)
        raise
    n += 1
    kinds['twisted' if twisted else 'plain'] = kinds.get('twisted' if twisted else 'plain', 0) + 1
    if time.time() - last > 30:
        print("  ... %d gradients so far (%s)" % (n, what), flush=True)
        last = time.time()
print("%d random gradients within 1e-9 of the oracle in %.0f s: %s" % (n, time.time() - t0, kinds))
