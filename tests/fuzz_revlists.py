"""Random genealogies through the device builders of the reverse pass's integer lists (phylo_revlists_dev.h) against the host
builders, entry by entry (tests/test_gpu_grad.py: _compare_device_lists_with_host).  python tests/fuzz_revlists.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylo_amd import _ffi               # noqa: E402
from tests import test_gpu_grad as T     # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
t0, n, last = time.time(), 0, time.time()
while time.time() - t0 < budget:
    N = int(rng.integers(2, 30))
    K = int(rng.choice([1, 2, 3, 63, 64, 65, 1000, 1024, 1025, 2048, 2500, 4096, 4097, 6000, 8192])) if rng.integers(0, 3) == 0 else int(rng.integers(1, 3000))
    if (N - 1) * K > 120000:
        K = max(1, 120000 // (N - 1))
    survivors = int(rng.choice([1, 2, 5, 40, max(1, K // 2), K]))
    what = "N=%d K=%d survivors=%d" % (N, K, survivors)
    try:
        anc, child = T._fast_genealogy(rng, N, K, survivors)
        with _ffi.Context(K, N, 4) as ctx:
            d = ctx.debug_device_lists(anc, child)
            d2 = ctx.debug_device_lists(anc, child)
        T._compare_device_lists_with_host(N, K, d, d2)
    except Exception:
        print("FAILED:", what, flush=True)
        raise
    n += 1
    if time.time() - last > 20:
        print("%d genealogies ok (last: %s)" % (n, what), flush=True)
        last = time.time()
print("fuzz_revlists: %d genealogies, device lists == host lists" % n)
