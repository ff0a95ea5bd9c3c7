"""Determinism soak of the reverse pass on one MI355X: primate.p K = 2048 and DS1 K = 4096, a few seeds, the kept graph's
gradient again and again -- the one-launch chains (completion words inside a launch) must give the same bits every time, and the
bits of the launch-per-rank-event form (PHYLO_GRAD_ROWS_CHAIN + PHYLO_GRAD_COEFF_CHAIN, in a child process).
python tools/soak_grad_determinism.py [repetitions]"""
import hashlib
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phylo_amd import _ffi, model as M  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
child = len(sys.argv) > 2 and sys.argv[2] == 'child'
cases = [('primate_data', 2048, reps), ('hohna_data_1', 4096, max(reps // 10, 5))]
digests = {}
t0 = time.time()
for ds, K, n in cases:
    g = load_dataset(ds)['genome']
    N, S, _ = g.shape
    Q = M.get_Q(M.init_y_q())
    pi = M.get_stationary_probs(np.zeros(4) + .25)
    lam = np.full(N - 1, 10.0)
    with _ffi.Context(K, N, S) as c:
        c.set_leaves(g)
        c.set_model(Q, pi, lam, lam)
        for seed in (0, 1, 2):
            c.sweep(seed, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
            first = None
            for i in range(1 if child else n):
                gr = c.sweep_backward()
                h = hashlib.sha256(b''.join(np.ascontiguousarray(gr[k]).tobytes() for k in ('d_lam_l', 'd_lam_r', 'd_pi', 'd_Q'))).hexdigest()
                if first is None:
                    first = h
                assert h == first, (ds, seed, i, "the gradient changed between repetitions")
            digests["%s/%d" % (ds, seed)] = first
if child:
    print(json.dumps(digests))
    sys.exit(0)
env = dict(os.environ, PHYLO_GRAD_ROWS_CHAIN='1', PHYLO_GRAD_COEFF_CHAIN='1')
out = subprocess.run([sys.executable, os.path.abspath(__file__), '1', 'child'], env=env, capture_output=True, text=True, check=True).stdout
chain = json.loads(out.strip().splitlines()[-1])
assert chain == digests, "one-launch chains and a launch per rank event differ"
print("reverse pass: %d + %d repetitions x 3 seeds, the same bits every time and in both forms (%.0f s)" % (cases[0][2], cases[1][2], time.time() - t0))
