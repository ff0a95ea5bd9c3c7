"""Out-of-band hand-off of the 128-byte RCCL unique id between the ranks of ONE node.

torch.distributed.run gives every rank MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE; its own store speaks a
private protocol, and this package does not import PyTorch, so the id travels through a file that rank 0
writes atomically under /tmp (single node by contract).  The directory name carries MASTER_PORT and the
launcher's pid (all ranks share one parent), so back-to-back runs on the same port cannot see stale ids.
"""
from __future__ import annotations

import os
import time


def _dir():
    tag = "%s_%s_%d" % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())
    return os.path.join(os.environ.get('PHYLO_RDZV_DIR', '/tmp'), "phylo_rdzv_" + tag)


_calls = 0


def exchange_comm_id(rank, world, make_id, timeout=300.0):
    """rank 0 calls make_id() and publishes it; every rank returns the same bytes.  Every call of a process uses its own file
    (all ranks make the same calls in the same order), so a second communicator of the same run never reads the first one's id."""
    global _calls
    d = _dir()
    path = os.path.join(d, "comm_id_%d.bin" % _calls)
    _calls += 1
    if rank == 0:
        cid = make_id()
        os.makedirs(d, exist_ok=True)
        tmp = path + ".tmp.%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(cid)
        os.replace(tmp, path)
        return cid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                cid = f.read()
            if len(cid) >= 128:
                return cid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL id from rank 0 at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.01)


class FileSync:
    """Barrier and all-gather of one float between the ranks of ONE node through files in the rendezvous directory: what bench.py
    falls back to when no communicator could be built (first contact with more than one GPU), so that the run still ends with a
    measured line.  Polling files costs ~0.1 ms of skew: callers time regions much longer than that."""

    def __init__(self, rank, world):
        self.rank, self.world, self.n = rank, world, 0
        self.dir = _dir()
        os.makedirs(self.dir, exist_ok=True)

    def allgather(self, value, timeout=300.0):
        tag = "fsync_%d" % self.n
        self.n += 1
        path = os.path.join(self.dir, "%s_r%d" % (tag, self.rank))
        with open(path + ".tmp", "w") as f:
            f.write(repr(float(value)))
        os.replace(path + ".tmp", path)
        out, t0 = [], time.time()
        for r in range(self.world):
            p = os.path.join(self.dir, "%s_r%d" % (tag, r))
            while True:
                try:
                    with open(p) as f:
                        out.append(float(f.read()))
                    break
                except (OSError, ValueError):
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rank %d: rank %d never reached %s" % (self.rank, r, tag))
                    time.sleep(0.0002)
        return out

    def barrier(self):
        self.allgather(0.0)

    def max(self, value):
        return max(self.allgather(value))
