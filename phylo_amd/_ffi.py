"""ctypes binding of libphylo_hip.so (include/phylo_hip.h).  No PyTorch, no CPU fallback.

The shared object is built in-tree by phylo_amd/csrc/build.sh (hipcc --offload-arch=gfx950).  Loading
fails loudly if it is missing; every compute call fails loudly (PhyloError) if no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libphylo_hip.so")

PHYLO_OK = 0
QUIRK_Q1_RAW_Q = 1 << 0
TWISTING = 1 << 1
TIME_KERNELS = 1 << 2
EAGER_NODES = 1 << 3
KEEP_GRAPH = 1 << 4
ONE_LAUNCH = 1 << 5
FLAGS_DEFAULT = QUIRK_Q1_RAW_Q
COMM_ID_BYTES = 128

EXPORTS = [
    "phylo_version", "phylo_last_error", "phylo_device_count", "phylo_create", "phylo_destroy",
    "phylo_set_leaves", "phylo_set_model", "phylo_expm_batched", "phylo_cond_likelihood_K",
    "phylo_forest_loglik", "phylo_tree_loglik", "phylo_resample", "phylo_log_zsmc", "phylo_sweep",
    "phylo_sweep_async", "phylo_sweep_batch_async", "phylo_sweep_batch_begin", "phylo_sweep_fetch_logz", "phylo_sweep_begin", "phylo_sweep_step", "phylo_sweep_step_a", "phylo_sweep_step_group", "phylo_sweep_finish", "phylo_sweep_fetch",
    "phylo_synchronize", "phylo_sweep_node", "phylo_sweep_backward",
    "phylo_math_probe", "phylo_debug_stamps", "phylo_debug_reverse_lists", "phylo_debug_device_lists", "phylo_debug_device_lists_of", "phylo_debug_remote_cache",
    "phylo_vi_gradients", "phylo_vi_apply",
    "phylo_site_tile", "phylo_set_site_tile", "phylo_get_site_tile",
    "phylo_comm_unique_id", "phylo_comm_init", "phylo_comm_share", "phylo_comm_allgather", "phylo_comm_max", "phylo_comm_barrier",
    "phylo_comm_exchange_kind",
]


class PhyloError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libphylo_hip error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("merge_ms", C.c_double), ("merge_launches", C.c_int32),
                ("n_launches", C.c_int32), ("units", C.c_double), ("alg_bytes", C.c_double)]


_lib = None


def load():
    """Load the library once; raises OSError with a build hint if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with phylo_amd/csrc/build.sh (needs hipcc); "
                          "there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.phylo_version.restype = C.c_char_p
        lib.phylo_last_error.restype = C.c_char_p
        lib.phylo_last_error.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def debug_reverse_lists(N, K, ancestors, child, early_free=True, rows_form=True, lookahead_nodes=None):
    """The host side of the reverse pass's integer lists on given ancestors [N-2][K] and children [N-1][K][2] (no GPU needed).
    Returns a dict of the arrays the device reads plus the per-rank-event offsets."""
    lib = load()
    R = N - 1
    nn = R * K
    cap = 2 * nn // 4 + 1
    n_lists = R * (K + 1) + 9 * nn + 1 + 2 * cap
    lists = np.zeros(n_lists, dtype=np.int32)
    meta = np.zeros(6 + 3 * (R + 1), dtype=np.int32)
    anc = None if R < 2 else np.ascontiguousarray(ancestors, dtype=np.int64)
    ch = np.ascontiguousarray(child, dtype=np.int32)
    la = None if lookahead_nodes is None or len(lookahead_nodes) == 0 else np.ascontiguousarray(lookahead_nodes, dtype=np.int32)
    rc = lib.phylo_debug_reverse_lists(C.c_int(N), C.c_int(K), _ptr(anc), _ptr(ch), C.c_int(int(early_free)), C.c_int(int(rows_form)),
                                       _ptr(la), C.c_int(0 if la is None else la.size), _ptr(lists), C.c_int64(n_lists), _ptr(meta),
                                       C.c_int(meta.size))
    if rc:
        raise PhyloError(rc, lib.phylo_last_error(None).decode())
    return _lists_dict(lists, meta, R, K)


def _lists_dict(lists, meta, R, K):
    nn = R * K
    cap = 2 * nn // 4 + 1
    out = {}
    o = 0
    for name, n in (("ad_off", R * (K + 1)), ("ad_idx", nn), ("par_off", nn + 1), ("par_idx", 2 * nn), ("heavy", nn), ("chunk_beg", cap),
                    ("chunk_cnt", cap), ("slow_flag", nn), ("slow_idx", nn), ("adp", nn)):
        out[name] = lists[o:o + n]
        o += n
    out["ad_off"] = out["ad_off"].reshape(R, K + 1)
    out["ad_idx"] = out["ad_idx"].reshape(R, K)
    for i, name in enumerate(("n_adp", "n_chunks", "max_chunks", "n_slow", "n_par", "cap")):
        out[name] = int(meta[i])
    out["ev_adp0"] = meta[6:6 + R + 1].copy()
    out["rank_chunk0"] = meta[6 + (R + 1):6 + 2 * (R + 1)].copy()
    out["ev_slow0"] = meta[6 + 2 * (R + 1):6 + 3 * (R + 1)].copy()
    return out


def vi_apply(N, jc, packed_vars, packed_grads, kind, lr, beta1=0.9, beta2=0.999, eps=1e-8, state=None):
    """phylo_vi_apply: the optimiser update on the packed variables IN PLACE (kind 0 gradient descent, 1 Adam with state =
    {'t': int, 'm': array, 'v': array}, updated in place too)."""
    lib = load()
    t = C.c_int64(0 if state is None else int(state['t']))
    m = None if state is None else state['m']
    v = None if state is None else state['v']
    rc = lib.phylo_vi_apply(C.c_int(N), C.c_int(int(jc)), _ptr(packed_vars), _ptr(packed_grads), C.c_int(kind), C.c_double(lr),
                            C.c_double(beta1), C.c_double(beta2), C.c_double(eps), C.byref(t), _ptr(m), _ptr(v))
    if rc:
        raise PhyloError(rc, lib.phylo_last_error(None).decode())
    if state is not None:
        state['t'] = t.value


def device_count():
    return int(load().phylo_device_count())


class Context:
    """Owns one phylo_ctx (one GPU).  Mirrors the C ABI one to one; numpy arrays in, numpy arrays out."""

    def __init__(self, K, N, S, A=4, device=0, flags=FLAGS_DEFAULT):
        self._lib = load()
        self._h = C.c_void_p()
        self.K, self.N, self.S, self.A = int(K), int(N), int(S), int(A)
        self.K_local, self.k0 = self.K, 0
        dev = (C.c_int * 1)(int(device))
        rc = self._lib.phylo_create(dev, C.c_int(1), C.c_int(self.K), C.c_int(self.N), C.c_int(self.S), C.c_int(self.A),
                                    C.c_uint32(flags), C.byref(self._h))
        if rc != PHYLO_OK:
            raise PhyloError(rc, (self._lib.phylo_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc != PHYLO_OK:
            raise PhyloError(rc, (self._lib.phylo_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.phylo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- state
    def set_site_tile(self, T):
        """contract v5: sites per tile of the canonical sum over sites (0 = the default phylo_site_tile(S))"""
        self._check(self._lib.phylo_set_site_tile(self._h, C.c_int(int(T))))

    def site_tile(self):
        return int(self._lib.phylo_get_site_tile(self._h))

    def set_leaves(self, genome_NxSxA):
        g = _f64(genome_NxSxA)
        if g.shape != (self.N, self.S, self.A):
            raise ValueError("genome shape %r != (%d, %d, %d)" % (g.shape, self.N, self.S, self.A))
        self._check(self._lib.phylo_set_leaves(self._h, _ptr(g)))

    def set_model(self, Q, pi, lam_l, lam_r, jc69_closed_form=False):
        Q, pi, ll, lr = _f64(Q), _f64(pi).reshape(-1), _f64(lam_l), _f64(lam_r)
        if Q.shape != (4, 4) or pi.shape != (4,) or ll.shape != (self.N - 1,) or lr.shape != (self.N - 1,):
            raise ValueError("bad model shapes")
        self._check(self._lib.phylo_set_model(self._h, _ptr(Q), _ptr(pi), _ptr(ll), _ptr(lr), C.c_int(int(jc69_closed_form))))

    # ---- ops
    def expm_batched(self, t):
        t = _f64(np.atleast_1d(t))
        P = np.empty((t.size, 4, 4))
        self._check(self._lib.phylo_expm_batched(self._h, _ptr(t), C.c_int(t.size), _ptr(P)))
        return P

    def cond_likelihood_K(self, l, r, tl, tr):
        l, r, tl, tr = _f64(l), _f64(r), _f64(tl), _f64(tr)
        if l.ndim != 3 or l.shape != r.shape or l.shape[2] != 4 or tl.shape != (l.shape[0],) or tr.shape != tl.shape:
            raise ValueError("bad shapes for cond_likelihood_K")
        out = np.empty_like(l)
        self._check(self._lib.phylo_cond_likelihood_K(self._h, _ptr(l), _ptr(r), _ptr(tl), _ptr(tr), C.c_int(l.shape[0]),
                                                      C.c_int(l.shape[1]), _ptr(out)))
        return out

    def forest_loglik(self, core_KxXxSx4, record_KxX):
        core = _f64(core_KxXxSx4)
        rec = np.ascontiguousarray(record_KxX, dtype=np.int32)
        if core.ndim != 4 or core.shape[3] != 4 or rec.shape != core.shape[:2]:
            raise ValueError("bad shapes for forest_loglik")
        K, X, S = core.shape[:3]
        out = np.empty(K)
        self._check(self._lib.phylo_forest_loglik(self._h, _ptr(core), _ptr(rec), C.c_int(K), C.c_int(X), C.c_int(S), _ptr(out)))
        return out

    def tree_loglik(self, left, right, bl, br, root, leaves, prior, want_root=True):
        leaves, prior = _f64(leaves), _f64(prior).reshape(-1)
        left = np.ascontiguousarray(left, dtype=np.int32)
        right = np.ascontiguousarray(right, dtype=np.int32)
        bl, br = _f64(bl), _f64(br)
        n_nodes, L, S = left.shape[0], leaves.shape[0], leaves.shape[1]
        out = C.c_double()
        rd = np.empty((S, 4)) if want_root else None
        self._check(self._lib.phylo_tree_loglik(self._h, C.c_int(n_nodes), C.c_int(L), C.c_int(S), _ptr(left), _ptr(right),
                                                _ptr(bl), _ptr(br), C.c_int(int(root)), _ptr(leaves), _ptr(prior),
                                                C.byref(out), _ptr(rd)))
        return out.value, rd

    def resample(self, logw, seed, step):
        w = _f64(logw).reshape(-1)
        idx = np.empty(w.size, dtype=np.int64)
        self._check(self._lib.phylo_resample(self._h, _ptr(w), C.c_int(w.size), C.c_uint64(seed), C.c_uint32(step), _ptr(idx)))
        return idx

    def log_zsmc(self, logw_RxK):
        w = _f64(logw_RxK)
        out = C.c_double()
        self._check(self._lib.phylo_log_zsmc(self._h, _ptr(w), C.c_int(w.shape[0]), C.c_int(w.shape[1]), C.byref(out)))
        return out.value

    def math_probe(self, op, x, y=None):
        x = _f64(x).reshape(-1)
        y = _f64(x if y is None else y).reshape(-1)
        out = np.empty_like(x)
        self._check(self._lib.phylo_math_probe(self._h, C.c_int(op), _ptr(x), _ptr(y), C.c_int(x.size), _ptr(out)))
        return out

    # ---- sweep
    def sweep_async(self, seed, flags=FLAGS_DEFAULT, M=1):
        self._check(self._lib.phylo_sweep_async(self._h, C.c_uint64(seed), C.c_uint32(flags), C.c_int(M)))

    def sweep_batch_async(self, seeds, flags=FLAGS_DEFAULT):
        """len(seeds) independent sweeps of K/len(seeds) particles each, in one set of launches."""
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        self._check(self._lib.phylo_sweep_batch_async(self._h, _ptr(sd), C.c_int(sd.size), C.c_uint32(flags)))

    def sweep_batch_begin(self, seeds, flags=FLAGS_DEFAULT):
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        self._check(self._lib.phylo_sweep_batch_begin(self._h, _ptr(sd), C.c_int(sd.size), C.c_uint32(flags)))

    def sweep_fetch_logz(self, G):
        out = np.empty(int(G))
        self._check(self._lib.phylo_sweep_fetch_logz(self._h, _ptr(out), C.c_int(int(G))))
        return out

    def sweep_begin(self, seed, flags=FLAGS_DEFAULT, M=1):
        self._check(self._lib.phylo_sweep_begin(self._h, C.c_uint64(seed), C.c_uint32(flags), C.c_int(M)))

    def sweep_step(self):
        self._check(self._lib.phylo_sweep_step(self._h))

    def sweep_step_a(self):
        self._check(self._lib.phylo_sweep_step_a(self._h))

    def sweep_finish(self):
        self._check(self._lib.phylo_sweep_finish(self._h))

    def synchronize(self):
        self._check(self._lib.phylo_synchronize(self._h))

    def sweep_fetch(self, arrays=True):
        R, K = self.N - 1, self.K_local
        out = {}
        if arrays:
            out = {'log_weights': np.empty((R, K)), 'log_likelihood': np.empty((R, K)),
                   'left_branches': np.empty((R, K)), 'right_branches': np.empty((R, K)),
                   'merges': np.empty((R, K, 2), dtype=np.int32),
                   'ancestors': np.empty((max(R - 1, 0), K), dtype=np.int64)}
        z = C.c_double()
        st = Stats()
        g = out.get
        self._check(self._lib.phylo_sweep_fetch(self._h, _ptr(g('log_weights')), _ptr(g('log_likelihood')),
                                                _ptr(g('left_branches')), _ptr(g('right_branches')), _ptr(g('merges')),
                                                _ptr(g('ancestors')), C.byref(z), C.byref(st)))
        out['logZ'] = z.value
        out['stats'] = {f: getattr(st, f) for f, _ in Stats._fields_}
        return out

    def sweep(self, seed, flags=FLAGS_DEFAULT, M=1):
        self.sweep_async(seed, flags, M)
        return self.sweep_fetch()

    def sweep_node(self, r, k):
        out = np.empty((self.S, 4))
        self._check(self._lib.phylo_sweep_node(self._h, C.c_int(r), C.c_int(k), _ptr(out)))
        return out

    def sweep_backward(self):
        """Gradient of logZ of the last sweep (run with KEEP_GRAPH) w.r.t. lam_l, lam_r, pi, Q (raw quantities)."""
        R = self.N - 1
        out = {'d_lam_l': np.empty(R), 'd_lam_r': np.empty(R), 'd_pi': np.empty(4), 'd_Q': np.empty((4, 4))}
        st = Stats()
        self._check(self._lib.phylo_sweep_backward(self._h, _ptr(out['d_lam_l']), _ptr(out['d_lam_r']), _ptr(out['d_pi']),
                                                   _ptr(out['d_Q']), C.byref(st)))
        out['backward_ms'] = st.sweep_ms
        out['backward_host_ms'] = st.merge_ms          # host time of the integer lists inside backward_ms (built, or waited for)
        out['backward_lists'] = 'device' if st.merge_launches else 'host'   # who built them (phylo_revlists_dev.h / phylo_revlists.h)
        out['backward_launches'] = st.n_launches
        return out

    def vi_gradients(self, seed, flags, M, jc, packed_vars):
        """The gradient half of a VI training step in the library (phylo_vi_gradients): packed_vars = a_l | a_r | y_q | y_station.
        Returns (logZ, grads packed alike, forward stats, backward stats)."""
        vars_ = _f64(packed_vars)
        grads = np.empty_like(vars_)
        z, fwd, bwd = C.c_double(), Stats(), Stats()
        self._check(self._lib.phylo_vi_gradients(self._h, C.c_uint64(seed), C.c_uint32(flags), C.c_int(M), C.c_int(int(jc)), _ptr(vars_),
                                                 C.byref(z), _ptr(grads), C.byref(fwd), C.byref(bwd)))
        return z.value, grads, fwd, bwd

    def debug_device_lists(self, ancestors=None, child=None):
        """The reverse pass's integer lists as the device kernels build them from the last (lazy, KEEP_GRAPH, plain proposal) sweep,
        or from the genealogy given (ancestors [N-2][K], child [N-1][K][2]; the context then needs a new sweep before the next
        sweep_backward): the dict of debug_reverse_lists plus 'ancestors' and 'child' they were built from."""
        R, K = self.N - 1, self.K
        nn = R * K
        cap = 2 * nn // 4 + 1
        n_lists = R * (K + 1) + 9 * nn + 1 + 2 * cap
        lists = np.zeros(n_lists, dtype=np.int32)
        meta = np.zeros(6 + 3 * (R + 1), dtype=np.int32)
        if child is not None:
            anc = np.ascontiguousarray(ancestors if R > 1 else np.zeros((0, K)), dtype=np.int64).reshape(max(R - 1, 0), K)
            child = np.ascontiguousarray(child, dtype=np.int32).reshape(R, K, 2)
            self._check(self._lib.phylo_debug_device_lists_of(self._h, _ptr(anc) if R > 1 else None, _ptr(child), _ptr(lists),
                                                              C.c_int64(n_lists), _ptr(meta), C.c_int(meta.size)))
        else:
            anc = np.zeros((max(R - 1, 0), K), dtype=np.int64)
            child = np.zeros((R, K, 2), dtype=np.int32)
            self._check(self._lib.phylo_debug_device_lists(self._h, _ptr(lists), C.c_int64(n_lists), _ptr(meta), C.c_int(meta.size),
                                                           _ptr(anc) if R > 1 else None, _ptr(child)))
        out = _lists_dict(lists, meta, R, K)
        out['ancestors'] = anc
        out['child'] = child
        return out

    def debug_stamps(self):
        """[N][16] s_memrealtime ticks (100 MHz) of workgroup 0 of the last one-launch sweep (PHYLO_PERSIST_STAMPS=1)."""
        out = np.zeros((self.N, 16), dtype=np.uint64)
        self._check(self._lib.phylo_debug_stamps(self._h, _ptr(out), C.c_int(out.size)))
        return out

    # ---- multi-GPU
    def comm_init(self, rank, world, comm_id):
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._check(self._lib.phylo_comm_init(self._h, C.c_int(rank), C.c_int(world), buf))
        self.K_local = self.K // world
        self.k0 = rank * self.K_local

    def comm_share(self, owner):
        """Join `owner`'s communicator (a further sweep in flight on the same rank); keep `owner` alive."""
        self._check(self._lib.phylo_comm_share(self._h, owner._h))
        self._owner = owner
        world = owner.K // owner.K_local
        self.K_local = self.K // world
        self.k0 = (owner.k0 // owner.K_local) * self.K_local

    def comm_allgather_columns(self, a):
        """a: [..., K_local] on every rank -> [..., K] (rank order = particle order).  Collective."""
        a = np.ascontiguousarray(a)
        world = self.K // self.K_local
        if world == 1:
            return a
        out = np.empty((world,) + a.shape, dtype=a.dtype)
        self._check(self._lib.phylo_comm_allgather(self._h, _ptr(a), C.c_size_t(a.nbytes), _ptr(out)))
        return np.concatenate(list(out), axis=-1)

    def comm_allgather_blob(self, a):
        """a: the same shape and dtype on every rank -> [world, ...] (one row per rank).  Collective."""
        a = np.ascontiguousarray(a)
        world = self.K // self.K_local
        if world == 1:
            return a[None]
        out = np.empty((world,) + a.shape, dtype=a.dtype)
        self._check(self._lib.phylo_comm_allgather(self._h, _ptr(a), C.c_size_t(a.nbytes), _ptr(out)))
        return out

    def debug_remote_cache(self):
        """(slots claimed by the last sweep, slots) of the local cache of remote nodes of a sharded context ((0, 0): no cache)"""
        used, cap = C.c_int(0), C.c_int(0)
        self._check(self._lib.phylo_debug_remote_cache(self._h, C.byref(used), C.byref(cap)))
        return used.value, cap.value

    def comm_exchange_kind(self):
        """'none' | 'rccl' | 'hostshm' | 'p2p': how the K-vectors of a rank event reach the other ranks"""
        return ('none', 'rccl', 'hostshm', 'p2p')[int(self._lib.phylo_comm_exchange_kind(self._h))]

    def comm_max(self, value):
        v = C.c_double(float(value))
        self._check(self._lib.phylo_comm_max(self._h, C.byref(v)))
        return v.value

    def comm_barrier(self):
        self._check(self._lib.phylo_comm_barrier(self._h))


def sweep_step_group(ctxs):
    """One rank event of several sweeps in flight (same rank event, one shared communicator): one grouped collective."""
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rc = load().phylo_sweep_step_group(arr, C.c_int(len(ctxs)))
    if rc != PHYLO_OK:
        raise PhyloError(rc, (load().phylo_last_error(None) or b"").decode())


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().phylo_comm_unique_id(buf)
    if rc != PHYLO_OK:
        raise PhyloError(rc, (load().phylo_last_error(None) or b"").decode())
    return buf.raw
