#!/usr/bin/env python3
"""Device time of the resampling scan + index search (phylo_resample) by K: the replicated scan of a sharded sweep runs on
K_total = n_gpus x K_local weights on every GPU.  Timed under rocprofv3 --kernel-trace (profiles/), or wall here."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from phylo_amd import _ffi, model as M  # noqa: E402

ctx = _ffi.Context(2, 3, 16)
ctx.set_leaves(np.ones((3, 16, 4)))
ctx.set_model(M.jc_Q(), np.full(4, 0.25), np.ones(2), np.ones(2))
rng = np.random.default_rng(0)
for K in (2048, 4096, 8192, 16384, 32768):
    lw = rng.normal(scale=30.0, size=K) - 6000.0
    for _ in range(3):
        ctx.resample(lw, 1, 1)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.resample(lw, 1, 1)
    print("K=%6d  phylo_resample wall %.1f us per call (H2D + scan + search + D2H)" % (K, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
ctx.close()
