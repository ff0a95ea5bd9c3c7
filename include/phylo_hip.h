/* phylo_hip.h -- C ABI of libphylo_hip.so: the MI355X-native Felsenstein-pruning likelihood and
 * CSMC particle loop of amoretti86/phylo (vcsmc.py / csmc.py), behind plain pointers and sizes.
 *
 * The reference has no FFI of its own: the path sits behind Python methods.  Each
 * entry point below names the reference method it stands in for (file:line into the reference);
 * the Python classes in phylo_amd/ (VCSMC, CSMC) keep the reference's names and argument meaning and
 * call these through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success, a negative PHYLO_E* code on failure; the message is
 *     available from phylo_last_error(ctx) (ctx may be NULL for failures of phylo_create);
 *   - the caller owns every host buffer (C-contiguous; double = IEEE binary64, int32/int64 as named);
 *     the library never keeps a host pointer past return;
 *   - the library owns all device memory and releases it in phylo_destroy;
 *   - a ctx is bound to ONE GPU and is not thread-safe; distinct ctxs are independent.  Multi-GPU =
 *     one process (rank) per GPU, joined by phylo_comm_init (RCCL over xGMI);
 *   - calls are synchronous at the boundary unless the name ends in _async;
 *   - no global RNG state: every stochastic entry point takes (seed, step) and follows the
 *     counter-based contract in DESIGN.md (Philox4x32-10).
 *   - there is no CPU fallback: without a usable HIP device every compute entry point fails with
 *     PHYLO_ENODEVICE.
 */
#ifndef PHYLO_HIP_H
#define PHYLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phylo_ctx phylo_ctx;

enum {
    PHYLO_OK = 0,
    PHYLO_EINVAL = -1,    /* bad argument (shape, NULL pointer, state not set) */
    PHYLO_ENODEVICE = -2, /* no HIP device / device id out of range */
    PHYLO_EHIP = -3,      /* a HIP runtime call failed */
    PHYLO_ENOMEM = -4,    /* device allocation failed */
    PHYLO_ECOMM = -5,     /* RCCL / multi-rank failure */
    PHYLO_ESTATE = -6     /* call order (e.g. sweep before set_leaves/set_model) */
};

/* sweep / model flags */
enum {
    PHYLO_QUIRK_Q1_RAW_Q = 1u << 0,    /* weight subtracts q = 1/C(n,2) itself, not log q (vcsmc.py:298,392).
                                          Set = as the reference.  */
    PHYLO_TWISTING = 1u << 1,          /* twisted/nested proposal of vncsmc.py:295-416 (uses M)          */
    PHYLO_TIME_KERNELS = 1u << 2,      /* bracket the dominant launch of every rank event with HIP events: the merge, or the
                                          look-ahead potentials of a twisted sweep (profiling runs; phylo_stats.merge_ms) */
    PHYLO_EAGER_NODES = 1u << 3,       /* always store every new node's partial likelihoods.  Default with the plain
                                          proposal (one GPU: always; sharded: S >= 8192): only nodes whose creator
                                          survives the next resampling are written (the rest are dead stores);
                                          results are identical either way.
                                          The nodes of the LAST rank event are never read by a merge and are not stored
                                          either (phylo_sweep_node writes them on demand) unless this flag is set */
    PHYLO_KEEP_GRAPH = 1u << 4,        /* keep what phylo_sweep_backward needs (root-table history of every rank
                                          event, every node; with PHYLO_TWISTING also every sub-sample's branch lengths,
                                          transition matrices and potential); one GPU.  Nodes are stored eagerly with
                                          PHYLO_TWISTING or more than 4096 sites; otherwise they stay lazy (the reverse pass then
                                          reads no node but the adopted ones) */
    PHYLO_ONE_LAUNCH = 1u << 5,        /* run the whole sweep as ONE launch of resident workgroups (phylo_persist.h) where that
                                          form applies (phylo_sweep_async / phylo_sweep_batch_async on one GPU, plain proposal,
                                          N <= 32, small nodes) instead of launches per rank event (scan, bookkeeping,
                                          materialise, merge).  Same bits either way; PHYLO_ONE_LAUNCH=1 in the environment of
                                          phylo_create sets it for every sweep of the context.  */
    PHYLO_FLAGS_DEFAULT = PHYLO_QUIRK_Q1_RAW_Q
};

typedef struct phylo_stats {
    double sweep_ms;        /* device time of the whole sweep (hipEvents on the ctx stream)            */
    double merge_ms;        /* sum of the launch durations of the rank events' dominant kernel (only with PHYLO_TIME_KERNELS):
                               the merge, or pk_twist_potentials for a twisted sweep.  From phylo_sweep_backward: the
                               host time spent building the integer lists                                  */
    int32_t merge_launches; /* number of launches in that sum                                          */
    int32_t n_launches;     /* kernel launches in the sweep                                            */
    double units;           /* particle-site-likelihoods computed by this rank: K_local * S * (N-1)    */
    double alg_bytes;       /* 96 B * units (two child reads + one parent write, fp64 x 4 states)      */
} phylo_stats;

const char* phylo_version(void);
const char* phylo_last_error(const phylo_ctx* ctx);

/* Number of visible HIP devices (0 if none / no driver).  Does not create a context. */
int phylo_device_count(void);

/* VCSMC.__init__ + the sizes of sample_phylogenies (vcsmc.py:110-118, 406-426): K particles (global
 * count), N taxa, S sites, A = 4 states.  device_ids/n_gpus: this build runs one GPU per process, so
 * n_gpus must be 1; more GPUs join through phylo_comm_init. */
int phylo_create(const int* device_ids, int n_gpus, int K, int N, int S, int A, uint32_t flags,
                 phylo_ctx** out);
int phylo_destroy(phylo_ctx* ctx);

/* The arithmetic contract's site tile (contract v5): sum_s log(pi . x[s]) of compute_forest_posterior
 * (vcsmc.py:240-242) is taken tile by tile -- T sites per tile, 64 log-product columns inside a tile, tile values added left to
 * right -- so that one wavefront owns a (row, tile).  phylo_site_tile(S) is the default T for rows of S sites (the CPU oracle
 * uses the same value); phylo_set_site_tile overrides it for this context (T a multiple of 64 in [64, 4096]; 0 = the default),
 * which changes results in the last bits only and drops the sweep state (call it before the first sweep; PHYLO_ESTATE once a
 * communicator is set).  PHYLO_SITE_TILE=T in the environment of phylo_create does the same.  phylo_get_site_tile returns the
 * context's T. */
int phylo_site_tile(int S);
int phylo_set_site_tile(phylo_ctx* ctx, int T);
int phylo_get_site_tile(const phylo_ctx* ctx);

/* datadict['genome'] [N,S,4] float64 (runner.py:107-115); stored once, not K-replicated
 * (the reference replicates it K-fold at vcsmc.py:479).  Returns when the caller's buffer has been read (small alignments are
 * staged in pinned memory and go up behind the call, ordered before every later call on the context). */
int phylo_set_leaves(phylo_ctx* ctx, const double* genome_NxSxA);

/* Model of VCSMC.__init__ / get_Q / get_stationary_probs (vcsmc.py:119-148), already evaluated by the
 * host: Q row-major 4x4, pi[4], lam_l / lam_r = exp(branch params) [N-1].  jc69_closed_form != 0 uses
 * P_ii = 1/4 + 3/4 e^-t for the JC69 Q instead of the generic Pade expm.  Returns when the 42 numbers are staged;
 * the upload is ordered before every later call on the context. */
int phylo_set_model(phylo_ctx* ctx, const double* Q16, const double* pi4, const double* lam_l,
                    const double* lam_r, int jc69_closed_form);

/* tf.linalg.expm(tensordot(t, Q, 0)) (vcsmc.py:181-184): P[i] = expm(Q * t[i]), [n,4,4]. */
int phylo_expm_batched(phylo_ctx* ctx, const double* t, int n, double* P_nx4x4);

/* VCSMC.broadcast_conditional_likelihood_K (vcsmc.py:180-188) == csmc.conditional_likelihood per
 * particle (csmc.py:300-309): out[k,s,:] = (l[k,s,:] @ P(tl[k])) * (r[k,s,:] @ P(tr[k])). */
int phylo_cond_likelihood_K(phylo_ctx* ctx, const double* l_KxSx4, const double* r_KxSx4,
                            const double* tl_K, const double* tr_K, int K, int S, double* out_KxSx4);

/* VCSMC.compute_forest_posterior (vcsmc.py:231-245): out[k] = sum_x sum_s log(pi . core[k,x,s,:])
 *   - sum_x log (2 max(record[k,x], 2) - 3)!! */
int phylo_forest_loglik(phylo_ctx* ctx, const double* core_KxXxSx4, const int32_t* record_KxX, int K,
                        int X, int S, double* out_K);

/* CSMC.compute_log_conditional_likelihood (csmc.py:318-326) on an explicit binary tree.  Nodes
 * 0..n_leaves-1 are leaves (rows of leaves_LxSx4); node i >= n_leaves has children left[i], right[i]
 * (already-numbered nodes < i or leaves) with branch lengths bl[i], br[i].  prior4 is csmc's
 * `self.prior`.  out_loglik = sum_s log(prior . data_root[s]); root_data_Sx4 may be NULL. */
int phylo_tree_loglik(phylo_ctx* ctx, int n_nodes, int n_leaves, int S, const int32_t* left,
                      const int32_t* right, const double* bl, const double* br, int root,
                      const double* leaves_LxSx4, const double* prior4, double* out_loglik,
                      double* root_data_Sx4);

/* VCSMC.resample's index draw (vcsmc.py:284-285) / CSMC.resample (csmc.py:218-228): K iid draws from
 * softmax(logw), by the integer-CDF contract.  idx_K[k] in [0,K). */
int phylo_resample(phylo_ctx* ctx, const double* logw_K, int K, uint64_t seed, uint32_t step,
                   int64_t* idx_K);

/* VCSMC.compute_log_ZSMC (vcsmc.py:270-277): sum_r logsumexp_k(logw[r,k] - log K). */
int phylo_log_zsmc(phylo_ctx* ctx, const double* logw_RxK, int R, int K, double* out);

/* VCSMC.sample_phylogenies (vcsmc.py:406-451): the N-1 rank events, device-resident.  Any output
 * pointer may be NULL.  Shapes (K = this rank's particles when sharded, see phylo_comm_init):
 *   log_weights, log_lik, lbranch, rbranch : [(N-1), K]   (rows 1..N-1 of the reference's tensors)
 *   merges    : [(N-1), K, 2]  root-table slots (left, right) coalesced at each rank event
 *   ancestors : [(N-2), K]     resampling indices drawn before rank events 1..N-2 (global indices)
 *   logZ      : scalar; perf : timing of this sweep
 * M is the number of sub-samples of the twisted proposal (ignored without PHYLO_TWISTING; 1 <= M <= 1024,
 * C(N,2)*M <= 2^20 with it; beyond 8192 sub-samples per particle their weights leave LDS).  With PHYLO_TWISTING the weight subtracts the normalised log-potential of the
 * chosen (pair, sub-sample) (vncsmc.py:315-316,491) and merges[r,k] = (r1 < r2). */
int phylo_sweep(phylo_ctx* ctx, uint64_t seed, uint32_t flags, int M, double* log_weights,
                double* log_lik, double* lbranch, double* rbranch, int32_t* merges, int64_t* ancestors,
                double* logZ, phylo_stats* perf);

/* Same sweep, left on the device (no host copies); phylo_sweep_fetch copies the last sweep's outputs. */
int phylo_sweep_async(phylo_ctx* ctx, uint64_t seed, uint32_t flags, int M);
int phylo_sweep_fetch(phylo_ctx* ctx, double* log_weights, double* log_lik, double* lbranch,
                      double* rbranch, int32_t* merges, int64_t* ancestors, double* logZ,
                      phylo_stats* perf);
int phylo_synchronize(phylo_ctx* ctx);

/* G INDEPENDENT sweeps in one set of launches (throughput form for callers that need many sweeps: minibatches,
 * replicates): the context's K particles are G groups of K/G; group g is exactly the sweep of K/G particles with
 * seeds[g] (own draws, own resampling, own log Z-hat).  Outputs of phylo_sweep_fetch hold group g in columns
 * [g K/G, (g+1) K/G) (ancestors index inside the group); phylo_sweep_fetch_logz returns the G estimates.
 * Plain proposal.  Sharded contexts too: the K = G * (K/G) particle indices are sharded by contiguous ranges as
 * always (a group may straddle ranks), one all-gather per rank event carries all G sweeps.
 * phylo_sweep_batch_begin + phylo_sweep_step(_group) + phylo_sweep_finish is the stepwise form. */
int phylo_sweep_batch_async(phylo_ctx* ctx, const uint64_t* seeds, int G, uint32_t flags);
int phylo_sweep_batch_begin(phylo_ctx* ctx, const uint64_t* seeds, int G, uint32_t flags);
int phylo_sweep_fetch_logz(phylo_ctx* ctx, double* logZ_G, int G);

/* The same sweep issued one rank event at a time: begin (draws, tables), N-1 x step, finish (log Z-hat).
 * phylo_sweep_async is exactly begin + steps + finish.  A caller that keeps several sweeps in flight on sharded
 * contexts interleaves them rank event by rank event (A0 B0 C0 A1 B1 C1 ...), so that the collectives of the
 * shared communicator (phylo_comm_share) are issued in one order on every rank while the kernels of the other
 * sweeps run underneath them. */
int phylo_sweep_begin(phylo_ctx* ctx, uint64_t seed, uint32_t flags, int M);
int phylo_sweep_step(phylo_ctx* ctx);
/* First half of the next rank event on a sharded context with lazy nodes (marking the adopted nodes, the owner's
 * writes, the collective that orders them before the merges); a no-op otherwise.  phylo_sweep_step runs it itself
 * when the caller has not: a caller with several contexts in flight issues the first halves of all of them before
 * the second halves, so that no context's merge waits behind another context's all-gather. */
int phylo_sweep_step_a(phylo_ctx* ctx);
/* One rank event of n sweeps that are at the same rank event and share one communicator: the kernels of each on its
 * own stream, then ONE grouped all-gather for all of them, then each sweep's scan. */
int phylo_sweep_step_group(phylo_ctx** ctxs, int n);
int phylo_sweep_finish(phylo_ctx* ctx);

/* Partial-likelihood vector of the node created at rank event r by particle slot k in the last sweep,
 * [S,4] (test surface for the merge kernel inside the sweep).  After a lazy sweep the missing nodes are
 * written first; when sharded that step is a collective: every rank must make the call. */
int phylo_sweep_node(phylo_ctx* ctx, int r, int k, double* out_Sx4);

/* Reverse pass of the last sweep (which must have run with PHYLO_KEEP_GRAPH): the gradient of log Z-hat with
 * respect to the raw model quantities, d_lam_l[N-1], d_lam_r[N-1], d_pi[4], d_Q[16] (row-major).
 * Replaces the TensorFlow autodiff behind optimizer.minimize(self.cost), vcsmc.py:488-491,534 (cost = -logZ):
 * resampling indices, pair picks and gather indices are constants, branch lengths are reparameterised samples
 * b = -log(U)/lambda (vcsmc.py:353-356), everything else is differentiated.  After a twisted sweep
 * (PHYLO_TWISTING | PHYLO_KEEP_GRAPH) that includes the look-ahead potentials of every (pair, sub-sample), whose normalised
 * value of the chosen one enters the weight (vncsmc.py:399-401, 491; no stop_gradient there).  With jc69_closed_form the Q and pi
 * outputs are still produced (the reference holds them constant; the host ignores them).
 * May be called right after phylo_sweep_async (before the fetch): it is then queued behind the sweep without a host round trip.
 * perf (may be NULL): sweep_ms = device time of the reverse pass, host step included; merge_ms = that host step (building the
 * integer lists of adopters and parents) alone; n_launches. */
int phylo_sweep_backward(phylo_ctx* ctx, double* d_lam_l, double* d_lam_r, double* d_pi, double* d_Q,
                         phylo_stats* perf);

/* The host half of a VI training step in the library (reference: optimizer.minimize(self.cost), vcsmc.py:488-491; the NumPy
 * statement of the same formulas is phylo_amd/train.py).  Variables packed as a_l[N-1] | a_r[N-1] | y_q[16] | y_station[4] (the
 * reference's 'left_branches_param', 'right_branches_param', 'Qmatrix', 'Stationary_probs': log-rates, vcsmc.py:119-124).
 * phylo_vi_gradients: model from the variables (vcsmc.py:133-148; jc != 0: the JC69 constants) -> phylo_set_model -> sweep with
 * PHYLO_KEEP_GRAPH on the context's leaves -> phylo_sweep_backward -> chain rules; grads = d logZ / d variables, packed alike
 * (zeros for y_q, y_station under JC69).  fwd / bwd (may be NULL): phylo_sweep_fetch's and phylo_sweep_backward's stats.
 * phylo_vi_apply: kind 0 tf.train.GradientDescentOptimizer (var += lr d logZ / d var), 1 tf.train.AdamOptimizer (TF 1.15
 * defaults are beta1 0.9, beta2 0.999, eps 1e-8; t, m, v: its state, m and v packed like the variables, zero at the start). */
int phylo_vi_gradients(phylo_ctx* ctx, uint64_t seed, uint32_t flags, int M, int jc, const double* vars, double* logZ, double* grads,
                       phylo_stats* fwd, phylo_stats* bwd);
int phylo_vi_apply(int n_taxa, int jc, double* vars, const double* grads, int kind, double lr, double beta1, double beta2, double eps,
                   int64_t* t, double* m, double* v);

/* Diagnostics of the one-launch sweep: with PHYLO_PERSIST_STAMPS=1 in the environment when the context is created,
 * workgroup 0 stamps s_memrealtime (100 MHz ticks) at its phase boundaries; out receives [N][16] values (rows 0..N-2: rank
 * events; row N-1: prologue).  No effect on results; not for timed runs. */
int phylo_debug_stamps(phylo_ctx* ctx, uint64_t* out, int n);

/* Test hook, no GPU needed: the host side of phylo_sweep_backward's integer lists (phylo_amd/csrc/phylo_revlists.h), run on
 * caller-supplied ancestors [N-2][K] (int64, as phylo_sweep returns them) and children [N-1][K][2] (node ids: leaf < N, else
 * N + r K + k of an EARLIER rank event: the children of rank event 0 are leaves and are not looked at).  lookahead_nodes: node ids with look-ahead entries (twisted proposal), may be NULL.  lists receives the slab the
 * device reads (ad_off | ad_idx | par_off | par_idx | heavy | chunk_beg | chunk_cnt | slow_flag | slow_idx | adp; R (K+1) + 9 R K
 * + 1 + 2 cap ints, cap = 2 R K / 4 + 1, R = N - 1); meta: n_adp, n_chunks, max_chunks, n_slow, n_par, cap, then ev_adp0[R+1],
 * rank_chunk0[R+1], ev_slow0[R+1].  tests/test_revlists_cpu.py checks it against a restatement in NumPy. */
int phylo_debug_reverse_lists(int N, int K, const int64_t* ancestors, const int32_t* child, int early_free, int rows_form,
                              const int32_t* lookahead_nodes, int n_lookahead, int32_t* lists, int64_t n_lists, int32_t* meta,
                              int n_meta);

/* The same lists built by the device kernels (phylo_revlists_dev.h) from the graph of the preceding lazy sweep with
 * PHYLO_KEEP_GRAPH, copied back in the same layout (what the builders do not write reads -1; heavy[] holds GLOBAL chunk indices,
 * rank_chunk0 in meta is zero), plus, when not NULL, the ancestors [N-2][K] and children [N-1][K][2] they were built from:
 * tests/test_gpu_grad.py compares them with phylo_debug_reverse_lists on those. */
int phylo_debug_device_lists(phylo_ctx* ctx, int32_t* lists, int64_t n_lists, int32_t* meta, int n_meta, int64_t* ancestors,
                             int32_t* child);
/* ... and from a genealogy the caller gives (the context's N and K; it replaces the last sweep's: sweep again before the next
 * phylo_sweep_backward): tests/test_gpu_grad.py replays the cases of tests/test_revlists_cpu.py through the device builders. */
int phylo_debug_device_lists_of(phylo_ctx* ctx, const int64_t* ancestors, const int32_t* child, int32_t* lists, int64_t n_lists,
                                int32_t* meta, int n_meta);

/* Bit-level probe of the device arithmetic contract: op 0 exp(x), 1 log(x), 2 x/y, 3 fma(x,y,x). */
int phylo_math_probe(phylo_ctx* ctx, int op, const double* x, const double* y, int n, double* out);

/* ---- multi-GPU: one process per GPU, particles sharded by contiguous ranges ------------------- */
#define PHYLO_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks out of band. */
int phylo_comm_unique_id(char id[PHYLO_COMM_ID_BYTES]);
/* Join `world` ranks.  The ctx must have been created with the GLOBAL K; afterwards this rank owns
 * particles [rank*K/world, (rank+1)*K/world) and sweep outputs are this shard's columns. */
int phylo_comm_init(phylo_ctx* ctx, int rank, int world, const char id[PHYLO_COMM_ID_BYTES]);
/* A further context of this process joins `owner`'s communicator (same rank, same world) instead of creating its
 * own: all collectives of the process then run on one stream of one communicator, in host issue order.  Collective
 * (peer pools are mapped); every rank makes the call for its contexts in the same order.  `owner` must outlive ctx. */
int phylo_comm_share(phylo_ctx* ctx, phylo_ctx* owner);
/* All-gather of a host blob of `bytes` bytes per rank (all = world * bytes, rank order): how a sharded caller
 * assembles per-particle outputs (ancestors, merges, branches) for host-side tree reconstruction. */
int phylo_comm_allgather(phylo_ctx* ctx, const void* mine, size_t bytes, void* all);
/* barrier + max over ranks of *value: an RCCL all-reduce (ncclMax) of one double on the communicator's stream (the
 * host-mediated test transport gathers and takes the max on the host); identity when no communicator is set. */
int phylo_comm_max(phylo_ctx* ctx, double* value);
int phylo_comm_barrier(phylo_ctx* ctx);
/* How this context exchanges the K-vectors of a rank event with the other ranks: 0 not sharded; 1 RCCL all-gather; 2 the
 * host-mediated test transport (PHYLO_COMM=hostshm); 3 the device-side exchange (every rank writes into the peers' hipIpc-mapped
 * slabs and raises a flag: no collective call per rank event; the default when sharded, PHYLO_P2P=0 turns it off). */
int phylo_comm_exchange_kind(const phylo_ctx* ctx);
/* Sharded contexts keep the remote nodes their particles merge in a local cache, fetched once per sweep over the peer mapping
 * (pk_pull_remote_children; PHYLO_NO_REMOTE_CACHE=1: every remote child is read in place, PHYLO_REMOTE_CACHE_CAP: slots).
 * used: slots claimed by the last sweep (more than cap: the rest was read in place); cap: slots (0: no cache). */
int phylo_debug_remote_cache(phylo_ctx* ctx, int* used, int* cap);

#ifdef __cplusplus
}
#endif
#endif /* PHYLO_HIP_H */
