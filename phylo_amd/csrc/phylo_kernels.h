// phylo_kernels.h -- HIP kernels of the CSMC hot path for gfx950 (MI355X).  See DESIGN.md for the
// data layout and the per-kernel roofline accounting.  All arithmetic follows phylo_math.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phylo_math.h"

#define PK_COLS 256          // canonical site-sum columns == threads per merge workgroup
#define PK_AUX 8             // per-particle scalars handed from the bookkeeping kernel to the merge epilogue
#define PK_MAX_TAXA 512

// aux slots
enum { AUX_SUM_REM = 0, AUX_FPRIOR, AUX_LPRIOR, AUX_RPRIOR, AUX_LL_TILDE, AUX_PAREN, AUX_LOGV, AUX_Q };

// ------------------------------------------------------------------------------------------------
// canonical sum over the 256 columns of a workgroup: adjacent-pair tree inside each 64-lane wave
// (xor butterfly 1,2,...,32: a+b == b+a bitwise, so every lane ends with the same value), then the four
// wave totals left to right.  oracle/csrc/oracle.c mirrors exactly this tree.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pk_block_canon_sum(double col, double* sh4) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) col = col + __shfl_xor(col, off, 64);
    __syncthreads();   // protect sh4 from a previous use
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = col;
    __syncthreads();
    return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}

__device__ __forceinline__ void pk_load4(const double* __restrict__ p, double* v) {
    const double2 a = reinterpret_cast<const double2*>(p)[0];
    const double2 b = reinterpret_cast<const double2*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

__device__ __forceinline__ void pk_store4(double* __restrict__ p, const double* v) {
    reinterpret_cast<double2*>(p)[0] = make_double2(v[0], v[1]);
    reinterpret_cast<double2*>(p)[1] = make_double2(v[2], v[3]);
}

// one site of broadcast_conditional_likelihood_K (vcsmc.py:185-187): out_j = (sum_i L_i Pl_ij)(sum_i R_i Pr_ij)
__device__ __forceinline__ void pk_merge_site(const double* L, const double* R, const double* Pl,
                                              const double* Pr, double* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double lp = L[0] * Pl[j];
        lp = pm_fma(L[1], Pl[4 + j], lp);
        lp = pm_fma(L[2], Pl[8 + j], lp);
        lp = pm_fma(L[3], Pl[12 + j], lp);
        double rp = R[0] * Pr[j];
        rp = pm_fma(R[1], Pr[4 + j], rp);
        rp = pm_fma(R[2], Pr[8 + j], rp);
        rp = pm_fma(R[3], Pr[12 + j], rp);
        out[j] = lp * rp;
    }
}

// pi . x  (compute_forest_posterior's matmul with the stationary vector, vcsmc.py:240)
__device__ __forceinline__ double pk_site_lik(const double* pi, const double* x) {
    double a = pi[0] * x[0];
    a = pm_fma(pi[1], x[1], a);
    a = pm_fma(pi[2], x[2], a);
    a = pm_fma(pi[3], x[3], a);
    return a;
}

// ------------------------------------------------------------------------------------------------
// k1: batched transition matrices
// ------------------------------------------------------------------------------------------------
__global__ void pk_expm_batched(const double* __restrict__ Q, const double* __restrict__ t, int n, int jc,
                                double* __restrict__ P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double q[16], p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = Q[j];
    if (jc) pm_jc69(t[i], p); else pm_expm4(q, t[i], p);
#pragma unroll
    for (int j = 0; j < 16; ++j) P[(size_t)i * 16 + j] = p[j];
}

// all branch lengths and transition matrices of one sweep: thread per (rank event r, local particle k).
// b = -log(U)/lambda_r (vcsmc.py:351-356); Pmat[r][k] = {P(b_l), P(b_r)}.
__global__ void pk_sweep_draws(const double* __restrict__ Q, const double* __restrict__ lam_l,
                               const double* __restrict__ lam_r, int jc, uint64_t seed, int R, int K, int k0,
                               double* __restrict__ bl, double* __restrict__ br, double* __restrict__ Pmat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * K) return;
    const int r = i / K, k = i - r * K;
    const pm_u32x4 x = pm_philox4x32((uint32_t)(k0 + k), (uint32_t)r, PM_STREAM_BRANCH, 0u, seed);
    const double tl = (-pm_log(pm_unit_oc(x.x, x.y))) / lam_l[r];
    const double tr = (-pm_log(pm_unit_oc(x.z, x.w))) / lam_r[r];
    bl[i] = tl;
    br[i] = tr;
    double q[16], p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = Q[j];
    double* out = Pmat + (size_t)i * 32;
    if (jc) pm_jc69(tl, p); else pm_expm4(q, tl, p);
#pragma unroll
    for (int j = 0; j < 16; ++j) out[j] = p[j];
    if (jc) pm_jc69(tr, p); else pm_expm4(q, tr, p);
#pragma unroll
    for (int j = 0; j < 16; ++j) out[16 + j] = p[j];
}

// ------------------------------------------------------------------------------------------------
// canonical sum_s log(pi . x[s]) of `rows` vectors [S,4]; one workgroup per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PK_COLS) void pk_row_loglik(const double* __restrict__ x, const double* __restrict__ pi4,
                                                          int S, double* __restrict__ out) {
    __shared__ double sh4[4];
    const double* row = x + (size_t)blockIdx.x * S * 4;
    double pi[4] = {pi4[0], pi4[1], pi4[2], pi4[3]};
    double col = 0.0;
    for (int s = threadIdx.x; s < S; s += PK_COLS) {
        double v[4];
        pk_load4(row + (size_t)s * 4, v);
        col = col + pm_log(pk_site_lik(pi, v));
    }
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// canonical sum of plain values: rows of length n
__global__ __launch_bounds__(PK_COLS) void pk_row_sum(const double* __restrict__ x, int n, double* __restrict__ out) {
    __shared__ double sh4[4];
    const double* row = x + (size_t)blockIdx.x * n;
    double col = 0.0;
    for (int s = threadIdx.x; s < n; s += PK_COLS) col = col + row[s];
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// forest posterior tail (vcsmc.py:242-245): out[k] = sum_x rowll[k,x] + sum_x -ldf[record[k,x]]
__global__ void pk_forest_tail(const double* __restrict__ rowll, const int32_t* __restrict__ record,
                               const double* __restrict__ ldf, int ldf_n, int K, int X, double* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double fl = 0.0, fp = 0.0;
    for (int x = 0; x < X; ++x) {
        fl = fl + rowll[(size_t)k * X + x];
        int c = record[(size_t)k * X + x];
        c = c < 0 ? 0 : (c > ldf_n ? ldf_n : c);
        fp = fp + (-ldf[c]);
    }
    out[k] = fl + fp;
}

// ------------------------------------------------------------------------------------------------
// k2 (API form): one workgroup per particle, explicit child arrays [K,S,4].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PK_COLS) void pk_merge_api(const double* __restrict__ l, const double* __restrict__ r,
                                                         const double* __restrict__ P /*[2K][16]: Pl then Pr*/,
                                                         int K, int S, double* __restrict__ out) {
    const int k = blockIdx.x;
    double Pl[16], Pr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        Pl[j] = P[(size_t)k * 16 + j];
        Pr[j] = P[(size_t)(K + k) * 16 + j];
    }
    const size_t base = (size_t)k * S * 4;
    for (int s = threadIdx.x; s < S; s += PK_COLS) {
        double L[4], R[4], o[4];
        pk_load4(l + base + (size_t)s * 4, L);
        pk_load4(r + base + (size_t)s * 4, R);
        pk_merge_site(L, R, Pl, Pr, o);
        pk_store4(out + base + (size_t)s * 4, o);
    }
}

// ------------------------------------------------------------------------------------------------
// a11: explicit-tree pruning.  Sites are independent through the whole tree: thread per site walks
// the post-order list; partials live in nodes[n_nodes][S][4]; P[2*i], P[2*i+1] for list entry i.
// ------------------------------------------------------------------------------------------------
__global__ void pk_tree_prune(double* __restrict__ nodes, const int32_t* __restrict__ order /*[n_int][3]: node,left,right*/,
                              int n_int, const double* __restrict__ P, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    for (int i = 0; i < n_int; ++i) {
        const int node = order[i * 3], lc = order[i * 3 + 1], rc = order[i * 3 + 2];
        double Pl[16], Pr[16], L[4], R[4], o[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            Pl[j] = P[(size_t)(2 * i) * 16 + j];
            Pr[j] = P[(size_t)(2 * i + 1) * 16 + j];
        }
        pk_load4(nodes + ((size_t)lc * S + s) * 4, L);
        pk_load4(nodes + ((size_t)rc * S + s) * 4, R);
        pk_merge_site(L, R, Pl, Pr, o);
        pk_store4(nodes + ((size_t)node * S + s) * 4, o);
    }
}

// ------------------------------------------------------------------------------------------------
// k5: resampling.  One workgroup of 1024 threads: max, integer weights, canonical fp sum (for the
// log-normaliser), inclusive integer prefix sum -> cdf[K] (uint64).  lse_out = m + log(sum) - log K.
// ------------------------------------------------------------------------------------------------
#define PK_SCAN_THREADS 1024
__global__ __launch_bounds__(PK_SCAN_THREADS) void pk_resample_scan(const double* __restrict__ logw, int K,
                                                                    uint64_t* __restrict__ cdf,
                                                                    double* __restrict__ lse_out) {
    __shared__ double shd[PK_SCAN_THREADS / 64];
    __shared__ uint64_t shu[PK_SCAN_THREADS / 64];
    __shared__ double sh4[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- max (NaN counts as -inf)
    double m = -pm_inf();
    for (int k = tid; k < K; k += PK_SCAN_THREADS) {
        const double v = logw[k];
        if (!pm_isnan(v) && v > m) m = v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) shd[wv] = m;
    __syncthreads();
    m = shd[0];
#pragma unroll
    for (int i = 1; i < PK_SCAN_THREADS / 64; ++i) m = shd[i] > m ? shd[i] : m;
    const bool all_bad = !(m > -pm_inf()) || m == pm_inf();
    // ---- canonical fp sum of the weights, 256 columns
    double col = 0.0;
    if (tid < PK_COLS) {
        for (int k = tid; k < K; k += PK_COLS) {
            const double v = logw[k];
            const double w = all_bad ? 1.0 : (pm_isnan(v) ? 0.0 : pm_exp(v - m));
            col = col + w;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) col = col + __shfl_xor(col, off, 64);
        if (lane == 0) sh4[wv] = col;
    }
    __syncthreads();
    if (tid == 0 && lse_out) {
        const double sum = ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
        const double mm = all_bad ? 0.0 : m;
        *lse_out = (mm + pm_log(sum)) - pm_log((double)K);
    }
    if (!cdf) return;
    // ---- integer inclusive scan; thread t owns the contiguous chunk [t*E, (t+1)*E)
    const int E = (K + PK_SCAN_THREADS - 1) / PK_SCAN_THREADS;
    const int lo = tid * E, hi = (lo + E < K) ? lo + E : K;
    uint64_t local = 0;
    for (int k = lo; k < hi; ++k) local += pm_weight_int(logw[k], m, all_bad);
    uint64_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) shu[wv] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (int i = 0; i < wv; ++i) wave_off += shu[i];
    uint64_t run = wave_off + incl - local;
    for (int k = lo; k < hi; ++k) {
        run += pm_weight_int(logw[k], m, all_bad);
        cdf[k] = run;
    }
}

__device__ __forceinline__ int pk_cdf_search(const uint64_t* __restrict__ cdf, int K, uint64_t thr) {
    int lo = 0, hi = K;             // first index with cdf[i] > thr
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > thr) hi = mid; else lo = mid + 1;
    }
    return lo < K ? lo : K - 1;
}

__global__ void pk_resample_search(const uint64_t* __restrict__ cdf, int K, int n_draw, int k0, uint64_t seed,
                                   uint32_t step, int64_t* __restrict__ idx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_draw) return;
    const pm_u32x4 x = pm_philox4x32((uint32_t)(k0 + k), step, PM_STREAM_RESAMPLE, 0u, seed);
    const uint64_t R = ((uint64_t)x.y << 32) | x.x;
    idx[k] = pk_cdf_search(cdf, K, pm_mulhi64(R, cdf[K - 1]));
}

// sum of per-rank log-normalisers, left to right (vcsmc.py:276 reduce_sum over rank events)
__global__ void pk_logz_total(const double* __restrict__ lse, int R, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double z = 0.0;
        for (int r = 0; r < R; ++r) z = z + lse[r];
        *out = z;
    }
}

// ------------------------------------------------------------------------------------------------
// sweep state
// ------------------------------------------------------------------------------------------------
__global__ void pk_init_tables(int32_t* __restrict__ roots, int32_t* __restrict__ cnt, int K, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * N) return;
    roots[i] = i % N;
    cnt[i] = 1;
}

struct pk_book_args {
    int r, n, N, K /*global*/, Kloc, k0;
    uint64_t seed;
    uint32_t flags;
    const int32_t* roots_old; const int32_t* cnt_old;     // [K global][N]
    int32_t* roots_new; int32_t* cnt_new;                 // [K global][N]
    const uint64_t* cdf;                                  // [K global] (r > 0)
    const double* ll_prev;                                // ll[r-1][K global] (r > 0)
    const double* nodell;                                 // per node id
    const double* ldf; int ldf_n;                         // -log (2 max(c,2) - 3)!! table by leaf count
    const double* bl; const double* br;                   // [R][Kloc] local slots
    double lam_l, lam_r, loglam_l, loglam_r, ll_tilde0;
    int32_t* child;                                       // [Kloc][2]
    double* aux;                                          // [Kloc][PK_AUX]
    int32_t* merges;                                      // [R][Kloc][2]
    int64_t* ancestors;                                   // [R-1][Kloc]
};

// Bookkeeping of one rank event for every particle of the GLOBAL population (thread per particle):
// resampling index (vcsmc.py:285), adoption of the ancestor's root table (the tf.gather of :286-288,
// on integer tables instead of partial likelihoods), uniform pair pick (:303-305), new root table
// (:361-373), and the scalar terms of the weight (:376-392) that do not depend on the new node.
// Integer state is replicated on every GPU; float outputs are written only for local slots.
__global__ void pk_rank_book(const pk_book_args a) {
    const int kg = blockIdx.x * blockDim.x + threadIdx.x;      // global particle
    if (kg >= a.K) return;
    const int n = a.n, N = a.N;
    int anc = kg;
    if (a.r > 0) {
        const pm_u32x4 x = pm_philox4x32((uint32_t)kg, (uint32_t)a.r, PM_STREAM_RESAMPLE, 0u, a.seed);
        const uint64_t R = ((uint64_t)x.y << 32) | x.x;
        anc = pk_cdf_search(a.cdf, a.K, pm_mulhi64(R, a.cdf[a.K - 1]));
    }
    const int32_t* ro = a.roots_old + (size_t)anc * N;
    const int32_t* co = a.cnt_old + (size_t)anc * N;
    // ---- keys of the n root slots
    uint32_t key[PK_MAX_TAXA];
    for (int b = 0; b < (n + 3) / 4; ++b) {
        const pm_u32x4 x = pm_philox4x32((uint32_t)kg, (uint32_t)a.r, PM_STREAM_PAIR, (uint32_t)b, a.seed);
        key[b * 4 + 0] = x.x; key[b * 4 + 1] = x.y; key[b * 4 + 2] = x.z; key[b * 4 + 3] = x.w;
    }
    int il = 0;                                     // largest key, lower slot on ties
    for (int i = 1; i < n; ++i) if (key[i] > key[il]) il = i;
    int ir = (il == 0) ? 1 : 0;                     // second largest
    for (int i = 0; i < n; ++i) if (i != il && i != ir && key[i] > key[ir]) ir = i;
    // the loop above keeps the lower slot on ties only if ir started at the lowest candidate: it does.
    const bool local = (kg >= a.k0) && (kg < a.k0 + a.Kloc);
    const int k = kg - a.k0;
    int32_t* rn = a.roots_new + (size_t)kg * N;
    int32_t* cn = a.cnt_new + (size_t)kg * N;
    // ---- remaining slots by ascending (key, slot); selection by repeated minimum
    double sum_rem = 0.0, fprior = 0.0;
    int vminus = 0;
    uint64_t last = 0;                              // (key << 32 | slot) + 1 of the previous pick; 0 = none
    for (int p = 0; p < n - 2; ++p) {
        uint64_t best = ~0ull;
        for (int i = 0; i < n; ++i) {
            if (i == il || i == ir) continue;
            const uint64_t c = (((uint64_t)key[i] << 32) | (uint32_t)i) + 1ull;
            if (c > last && c < best) best = c;
        }
        last = best;
        const int slot = (int)((best - 1ull) & 0xffffffffull);
        const int node = ro[slot];
        const int c = co[slot];
        rn[p] = node;
        cn[p] = c;
        if (local) sum_rem = sum_rem + a.nodell[node];
        fprior = fprior + (-a.ldf[c < a.ldf_n ? c : a.ldf_n]);
        vminus += c - (c == 1 ? 1 : 0);
    }
    const int cl = ro[il], cr = ro[ir];
    const int cnew = co[il] + co[ir];
    rn[n - 2] = a.N + a.r * a.K + kg;               // id of the node this particle creates now
    cn[n - 2] = cnew;
    if (!local) return;
    fprior = fprior + (-a.ldf[cnew < a.ldf_n ? cnew : a.ldf_n]);
    vminus += cnew - (cnew == 1 ? 1 : 0);
    a.child[k * 2 + 0] = cl;
    a.child[k * 2 + 1] = cr;
    a.merges[((size_t)a.r * a.Kloc + k) * 2 + 0] = il;
    a.merges[((size_t)a.r * a.Kloc + k) * 2 + 1] = ir;
    if (a.r > 0) a.ancestors[(size_t)(a.r - 1) * a.Kloc + k] = anc;
    // ---- branch-length log-priors over the slot-attached history rows 0..r with THIS rank's rate (Q3)
    double lp = 0.0, rp = 0.0;
    for (int j = 0; j <= a.r; ++j) {
        lp = lp + ((-a.lam_l) * a.bl[(size_t)j * a.Kloc + k] + a.loglam_l);
        rp = rp + ((-a.lam_r) * a.br[(size_t)j * a.Kloc + k] + a.loglam_r);
    }
    const double b_l = a.bl[(size_t)a.r * a.Kloc + k], b_r = a.br[(size_t)a.r * a.Kloc + k];
    const double paren = ((a.loglam_l - a.lam_l * b_l) + a.loglam_r) - a.lam_r * b_r;
    const double q = 1.0 / ((double)((n - 1) * n) / 2.0);      // 1 / ncr(n, 2), vcsmc.py:298
    double* ax = a.aux + (size_t)k * PK_AUX;
    ax[AUX_SUM_REM] = sum_rem;
    ax[AUX_FPRIOR] = fprior;
    ax[AUX_LPRIOR] = lp;
    ax[AUX_RPRIOR] = rp;
    ax[AUX_LL_TILDE] = (a.r > 0) ? a.ll_prev[anc] : a.ll_tilde0;
    ax[AUX_PAREN] = paren;
    ax[AUX_LOGV] = pm_log((double)vminus);
    ax[AUX_Q] = (a.flags & 1u) ? q : pm_log(q);
}

// ------------------------------------------------------------------------------------------------
// k2 + k3 + k8 (sweep form): one workgroup per local particle.  Reads the two child partials by node
// id (leaves or pool), writes the new node's partial, reduces sum_s log(pi . new[s]) canonically, and
// finishes log_likelihood_r and log w_r (vcsmc.py:376-392).
//   algorithmic traffic: 2 x 32 B read + 32 B written per (particle, site)  = 96 B / unit.
// ------------------------------------------------------------------------------------------------
struct pk_merge_args {
    const double* leaves;     // [N][S][4]
    double* pool;             // [(N-1)][Kloc][S][4] local nodes
    const int32_t* child;     // [Kloc][2]
    const double* Pmat;       // [Kloc][32] of this rank
    const double* pi;
    double* nodell;           // per node id (global ids)
    const double* aux;        // [Kloc][PK_AUX]
    double* logw_r;           // [Kloc]
    double* ll_r;             // [Kloc]
    int N, S, r, K /*global*/, Kloc, k0;
};

__global__ __launch_bounds__(PK_COLS) void pk_rank_merge(const pk_merge_args a) {
    __shared__ double sh4[4];
    const int k = blockIdx.x;
    const int cl = a.child[k * 2], cr = a.child[k * 2 + 1];
    const size_t node_sz = (size_t)a.S * 4;
    // a child is a leaf (id < N) or a node of the local pool: id = N + rho*K + kappa
    const double* Lp = cl < a.N ? a.leaves + (size_t)cl * node_sz
                                : a.pool + ((size_t)((cl - a.N) / a.K) * a.Kloc + ((cl - a.N) % a.K - a.k0)) * node_sz;
    const double* Rp = cr < a.N ? a.leaves + (size_t)cr * node_sz
                                : a.pool + ((size_t)((cr - a.N) / a.K) * a.Kloc + ((cr - a.N) % a.K - a.k0)) * node_sz;
    double* out = a.pool + ((size_t)a.r * a.Kloc + k) * node_sz;
    double Pl[16], Pr[16];
    const double* P = a.Pmat + (size_t)k * 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) { Pl[j] = P[j]; Pr[j] = P[16 + j]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double col = 0.0;
    for (int s = threadIdx.x; s < a.S; s += PK_COLS) {
        double L[4], R[4], o[4];
        pk_load4(Lp + (size_t)s * 4, L);
        pk_load4(Rp + (size_t)s * 4, R);
        pk_merge_site(L, R, Pl, Pr, o);
        pk_store4(out + (size_t)s * 4, o);
        col = col + pm_log(pk_site_lik(pi, o));
    }
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) {
        const double* ax = a.aux + (size_t)k * PK_AUX;
        a.nodell[a.N + a.r * a.K + a.k0 + k] = tot;
        const double fl = ax[AUX_SUM_REM] + tot;
        const double ll = ((fl + ax[AUX_FPRIOR]) + ax[AUX_LPRIOR]) + ax[AUX_RPRIOR];
        const double lw = (((ll - ax[AUX_LL_TILDE]) - ax[AUX_PAREN]) + ax[AUX_LOGV]) - ax[AUX_Q];
        a.ll_r[k] = ll;
        a.logw_r[k] = lw;
    }
}

// arithmetic probe
__global__ void pk_math_probe(int op, const double* __restrict__ x, const double* __restrict__ y, int n,
                              double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v;
    switch (op) {
        case 0: v = pm_exp(x[i]); break;
        case 1: v = pm_log(x[i]); break;
        case 2: v = x[i] / y[i]; break;
        default: v = pm_fma(x[i], y[i], x[i]); break;
    }
    out[i] = v;
}
