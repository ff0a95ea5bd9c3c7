"""ctypes wrapper of the C oracle (oracle/csrc/oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call([os.path.join(_HERE, "build.sh")], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ora_log_zsmc.restype = C.c_double
        # OpenMP sizes its team by the host's hardware threads; a container or GPU box grants far fewer (16 there), and an
        # oversubscribed team makes every small parallel region cost milliseconds.  Callers may still set_threads().
        if "OMP_NUM_THREADS" not in os.environ:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            _LIB.ora_set_threads(max(1, min(avail, 16)))
    return _LIB


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return int(lib().ora_num_threads())


def set_threads(n):
    lib().ora_set_threads(int(n))


def set_site_tile(T):
    """contract v5: site tile of the canonical sum over sites (0 = the policy of ora_site_tile)"""
    lib().ora_set_site_tile(C.c_int(int(T)))


def site_tile(S):
    return int(lib().ora_get_site_tile(C.c_long(int(S))))


def math_probe(op, x, y=None):
    x = _c(x)
    y = _c(x if y is None else y)
    out = np.empty_like(x)
    lib().ora_math_probe(C.c_int(op), _opt(x), _opt(y), C.c_int(x.size), _opt(out))
    return out


def philox(c0, c1, c2, c3, seed):
    out = (C.c_uint32 * 4)()
    lib().ora_philox4x32(C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), C.c_uint64(seed), out)
    return [int(v) for v in out]


def expm_batched(Q, t, jc=False):
    Q, t = _c(Q), _c(np.atleast_1d(t))
    P = np.empty((t.size, 4, 4))
    lib().ora_expm_batched(_opt(Q), _opt(t), C.c_int(t.size), C.c_int(int(jc)), _opt(P))
    return P


def cond_likelihood_K(Q, l, r, tl, tr, jc=False):
    Q, l, r, tl, tr = _c(Q), _c(l), _c(r), _c(tl), _c(tr)
    K, S = l.shape[0], l.shape[1]
    out = np.empty_like(l)
    lib().ora_cond_likelihood_K(_opt(Q), C.c_int(int(jc)), _opt(l), _opt(r), _opt(tl), _opt(tr), C.c_int(K), C.c_int(S),
                                _opt(out))
    return out


def forest_loglik(pi, core, record):
    pi, core, record = _c(pi).reshape(-1), _c(core), _c(record, np.int32)
    K, X, S = core.shape[0], core.shape[1], core.shape[2]
    out = np.empty(K)
    lib().ora_forest_loglik(_opt(pi), _opt(core), _opt(record), C.c_int(K), C.c_int(X), C.c_int(S), _opt(out))
    return out


def tree_loglik(Q, prior, left, right, bl, br, root, leaves, jc=False):
    Q, prior, leaves = _c(Q), _c(prior), _c(leaves)
    left, right, bl, br = _c(left, np.int32), _c(right, np.int32), _c(bl), _c(br)
    n_nodes, L, S = left.shape[0], leaves.shape[0], leaves.shape[1]
    out = C.c_double()
    rd = np.empty((S, 4))
    rc = lib().ora_tree_loglik(_opt(Q), C.c_int(int(jc)), C.c_int(n_nodes), C.c_int(L), C.c_int(S), _opt(left), _opt(right),
                               _opt(bl), _opt(br), C.c_int(int(root)), _opt(leaves), _opt(prior), C.byref(out), _opt(rd))
    if rc != 0:
        raise MemoryError("ora_tree_loglik")
    return out.value, rd


def resample(logw, seed, step):
    logw = _c(logw)
    idx = np.empty(logw.size, dtype=np.int64)
    lib().ora_resample(_opt(logw), C.c_int(logw.size), C.c_uint64(seed), C.c_uint32(step), _opt(idx))
    return idx


def log_zsmc(logw_RxK):
    w = _c(logw_RxK)
    return float(lib().ora_log_zsmc(_opt(w), C.c_int(w.shape[0]), C.c_int(w.shape[1])))


def sweep(genome, Q, pi, lam_l, lam_r, K, seed, flags=1, jc=False, want_nodes=False):
    genome, Q, pi, lam_l, lam_r = _c(genome), _c(Q), _c(pi).reshape(-1), _c(lam_l), _c(lam_r)
    N, S = genome.shape[0], genome.shape[1]
    R = N - 1
    out = {
        'log_weights': np.empty((R, K)), 'log_likelihood': np.empty((R, K)),
        'left_branches': np.empty((R, K)), 'right_branches': np.empty((R, K)),
        'merges': np.empty((R, K, 2), dtype=np.int32), 'ancestors': np.empty((max(R - 1, 0), K), dtype=np.int64),
    }
    nodes = np.empty((R, K, S, 4)) if want_nodes else None
    z = C.c_double()
    rc = lib().ora_sweep(_opt(genome), _opt(Q), _opt(pi), _opt(lam_l), _opt(lam_r), C.c_int(int(jc)), C.c_int(K), C.c_int(N),
                         C.c_int(S), C.c_uint64(seed), C.c_uint32(flags), _opt(out['log_weights']),
                         _opt(out['log_likelihood']), _opt(out['left_branches']), _opt(out['right_branches']),
                         _opt(out['merges']), _opt(out['ancestors']), C.byref(z), _opt(nodes))
    if rc != 0:
        raise MemoryError("ora_sweep")
    out['logZ'] = z.value
    if want_nodes:
        out['nodes'] = nodes
    return out


def sweep_twisted(genome, Q, pi, lam_l, lam_r, K, M, seed, jc=False, want_nodes=False, want_potentials=False):
    genome, Q, pi, lam_l, lam_r = _c(genome), _c(Q), _c(pi).reshape(-1), _c(lam_l), _c(lam_r)
    N, S = genome.shape[0], genome.shape[1]
    R = N - 1
    out = {
        'log_weights': np.empty((R, K)), 'log_likelihood': np.empty((R, K)),
        'left_branches': np.empty((R, K)), 'right_branches': np.empty((R, K)),
        'merges': np.empty((R, K, 2), dtype=np.int32), 'ancestors': np.empty((max(R - 1, 0), K), dtype=np.int64),
    }
    nodes = np.empty((R, K, S, 4)) if want_nodes else None
    pots = np.empty((R, K, (N * (N - 1) // 2) * M)) if want_potentials else None
    z = C.c_double()
    rc = lib().ora_sweep_twisted(_opt(genome), _opt(Q), _opt(pi), _opt(lam_l), _opt(lam_r), C.c_int(int(jc)), C.c_int(K),
                                 C.c_int(N), C.c_int(S), C.c_int(M), C.c_uint64(seed), _opt(out['log_weights']),
                                 _opt(out['log_likelihood']), _opt(out['left_branches']), _opt(out['right_branches']),
                                 _opt(out['merges']), _opt(out['ancestors']), C.byref(z), _opt(nodes), _opt(pots))
    if rc != 0:
        raise MemoryError("ora_sweep_twisted")
    out['logZ'] = z.value
    if want_nodes:
        out['nodes'] = nodes
    if want_potentials:
        out['potentials'] = pots
    return out
