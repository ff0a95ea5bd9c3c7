"""A/B of the reverse pass's switches on the VI training step (bench.py's vi_step): device-built lists against the host builders
(PHYLO_REV_HOST_LISTS), one stream against two.  python tools/vi_step_ab.py [dataset K] ..."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from phylo_amd.datasets import load_dataset  # noqa: E402

cases = [('primate_data', 2048), ('hohna_data_1', 4096)]
if len(sys.argv) > 2:
    cases = [(sys.argv[i], int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
switches = [(), ('PHYLO_REV_HOST_LISTS',), ('PHYLO_GRAD_ONE_STREAM',), ('PHYLO_GRAD_TWO_STREAMS',),
            ('PHYLO_REV_HOST_LISTS', 'PHYLO_GRAD_ONE_STREAM')]
if os.environ.get('AB_SWITCHES'):
    switches = [tuple(x for x in grp.split(',') if x) for grp in os.environ['AB_SWITCHES'].split(';')]
for ds, K in cases:
    g = load_dataset(ds)['genome']
    for sw in switches:
        for e in sw:
            os.environ[e.split('=')[0]] = e.split('=')[1] if '=' in e else '1'
        try:
            out = bench.vi_step_timing(g, K, steps=20)
        finally:
            for e in sw:
                del os.environ[e.split('=')[0]]
        print(json.dumps({"dataset": ds, "K": K, "switches": list(sw), "plain": out["plain"], "twisted_M1": out["twisted_M1"]}), flush=True)
