"""Plain-proposal VI steps for a kernel trace (rocprofv3 --kernel-trace): python tools/vi_step_trace.py [dataset K steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylo_amd import train as T  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402

ds = sys.argv[1] if len(sys.argv) > 1 else 'primate_data'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
g = load_dataset(ds)['genome']
N, S, _ = g.shape
v = T.Variables(N, np.log(10.0), False)
tr = T.Trainer(g, K, v, T.make_optimizer('Adam', 0.01), S, device=0, nested=False, M=1)
for i in range(steps):
    tr.step(np.arange(S), seed=i)
print(tr.last['raw']['forward_ms'], tr.last['raw']['backward_ms'], tr.last['raw'].get('backward_host_ms'))
tr.close()
