"""Host-side statement of the RNG contract (DESIGN.md "RNG contract"): Philox4x32-10, key = 64-bit seed,
counter = (particle, rank event, stream, block).  Used by the Python surface for the integer bookkeeping the
reference does in TensorFlow string/int ops (jump chains, extend_partial_state); the device draws the same
numbers in phylo_kernels.h."""
from __future__ import annotations

import numpy as np

STREAM_PAIR, STREAM_BRANCH, STREAM_RESAMPLE, STREAM_TWIST = 0, 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32(c0, c1, c2, c3, seed):
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = (x.copy() for x in np.broadcast_arrays(*c))
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = (p1 >> _S32) ^ c1 ^ np.uint64(k0), p1 & _MASK, (p0 >> _S32) ^ c3 ^ np.uint64(k1), p0 & _MASK
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def pair_order(K, n, seed, step, k0=0):
    """Uniform pair pick of vcsmc.py:303-305 under the contract: (coalesced [K,2], remaining [K,n-2])."""
    nb = (n + 3) // 4
    x = philox4x32(np.arange(k0, k0 + K)[:, None], step, STREAM_PAIR, np.arange(nb)[None, :], seed)
    keys = np.stack(x, axis=-1).reshape(K, nb * 4)[:, :n].astype(np.int64)
    slot = np.arange(n)[None, :]
    desc = np.argsort(-(keys << 10) + slot, axis=1, kind='stable')      # key descending, lower slot first
    coalesced = desc[:, :2].astype(np.int32)
    asc = (keys << 10) + slot
    np.put_along_axis(asc, coalesced.astype(np.int64), np.iinfo(np.int64).max, axis=1)
    remaining = np.argsort(asc, axis=1, kind='stable')[:, :n - 2].astype(np.int32)
    return coalesced, remaining
