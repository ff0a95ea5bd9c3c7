#!/usr/bin/env python3
"""One-launch sweep (phylo_persist.h) against the launch-per-rank-event path and the C oracle: bits and device time.
usage: python tests/probe_persist.py [K] [dataset] [reps]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import c_oracle as CO  # noqa: E402
from phylo_amd import _ffi, model as M  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dataset = sys.argv[2] if len(sys.argv) > 2 else 'primate_data'
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    g = load_dataset(dataset)['genome']
    N, S, _ = g.shape
    Q = M.get_Q(M.init_y_q())
    pi = M.get_stationary_probs(np.zeros(4) + 0.25)
    lam = np.full(N - 1, 10.0)
    ctx = _ffi.Context(K, N, S)
    ctx.set_leaves(g)
    ctx.set_model(Q, pi, lam, lam)
    for name, fl in (('one launch', _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH), ('per rank event', _ffi.FLAGS_DEFAULT)):
        out = ctx.sweep(0, flags=fl)
        ms = []
        for s in range(reps):
            ctx.sweep_async(s, flags=fl)
            ms.append(ctx.sweep_fetch(arrays=False)['stats']['sweep_ms'])
        t0 = time.perf_counter()
        for s in range(reps):
            ctx.sweep_async(s, flags=fl)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        print("%-15s launches=%d  device ms median %.4f min %.4f max %.4f | back-to-back wall %.4f ms/sweep | logZ %.6f"
              % (name, out['stats']['n_launches'], np.median(ms), min(ms), max(ms), wall, out['logZ']), flush=True)
        if name == 'one launch':
            first = out
        else:
            assert np.array_equal(first['log_weights'].view(np.uint64), out['log_weights'].view(np.uint64)), "paths differ"
            assert np.array_equal(first['ancestors'], out['ancestors'])
    if os.environ.get('PHYLO_PERSIST_STAMPS'):
        ctx.sweep(1)
        st = ctx.debug_stamps().astype(np.int64)
        print("phase stamps of workgroup 0 / thread 0 (its LAST particle of the rank event for the per-particle stamps), us:")
        print("  prologue %.2f us; shader clock over the kernel: %.3f GHz" % ((st[N - 1, 1] - st[N - 1, 0]) * 0.01,
              (st[N - 1, 5] - st[N - 1, 4]) / max(1, (st[N - 1, 6] - st[N - 1, 0])) * 0.1))
        cols = [('wait', 0, 1), ('scan', 1, 2), ('adopt', 2, 3), ('mater', 3, 4), ('particles', 4, 6), ('arrive', 6, 7),
                ('|B:rows', 8, 9), ('P', 9, 10), ('merge', 10, 11), ('epi', 11, 12), ('|partA', 13, 14)]
        print("  r  " + " ".join("%9s" % c[0] for c in cols) + "   | event")
        tot = np.zeros(len(cols))
        for r in range(N - 1):
            d = []
            for name, i0, i1 in cols:
                if r == 0 and i1 <= 4:
                    d.append(0.0)
                else:
                    d.append((st[r, i1] - st[r, i0]) * 0.01)
            tot += d
            print("  %2d " % r + " ".join("%9.2f" % x for x in d) + "   | %.2f" % ((st[r, 7] - st[r, 0]) * 0.01))
        print("  sum" + " ".join("%9.2f" % x for x in tot) + "   | kernel start to last arrival %.2f" % ((st[N - 2, 7] - st[N - 1, 0]) * 0.01))
    if K <= 4096:
        ref = CO.sweep(g, Q, pi, lam, lam, K, 0)
        assert np.array_equal(first['ancestors'], ref['ancestors'])
        assert np.array_equal(first['log_weights'].view(np.uint64), ref['log_weights'].view(np.uint64))
        assert first['logZ'] == ref['logZ']
        print("bit-exact vs the C oracle (seed 0)")
    ctx.close()


if __name__ == '__main__':
    main()
