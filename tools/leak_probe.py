"""Create and destroy contexts that use every mode; VRAM in use after n and 2n iterations (a leak grows linearly).
python tools/leak_probe.py [n]"""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylo_amd import _ffi, model as M          # noqa: E402
from phylo_amd.datasets import load_dataset     # noqa: E402

g = load_dataset("primate_data")["genome"][:, :200]
N, S, _ = g.shape
Q = M.get_Q(M.init_y_q())
pi = M.get_stationary_probs(np.zeros(4) + 0.25)
lam = np.full(N - 1, 10.0)


def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True).stdout
    d = json.loads(out)
    return int(d[list(d.keys())[0]]["VRAM Total Used Memory (B)"]) / 2 ** 20


def cycle(i):
    with _ffi.Context(256 + 4 * (i % 8), N, S) as c:
        c.set_leaves(g)
        c.set_model(Q, pi, lam, lam)
        c.sweep(i)
        c.sweep(i, _ffi.FLAGS_DEFAULT | _ffi.ONE_LAUNCH)
        c.sweep(i, _ffi.FLAGS_DEFAULT | _ffi.KEEP_GRAPH)
        c.sweep_backward()
        c.sweep(i, _ffi.FLAGS_DEFAULT | _ffi.TWISTING | _ffi.KEEP_GRAPH, 2)
        c.sweep_backward()
        c.sweep_batch_async([1, 2, 3, 4])
        c.sweep_fetch()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
marks = [used()]
for rep in range(3):
    for i in range(n):
        cycle(i)
    marks.append(used())
print("VRAM MiB in use: start %.0f, after %d / %d / %d cycles: %.0f / %.0f / %.0f" % (marks[0], n, 2 * n, 3 * n, marks[1], marks[2], marks[3]))
