// Reverse pass of the sweep: d log Z-hat / d(lam_l, lam_r, pi, Q)  (gfx950).
//
// The reference obtains this derivative by TensorFlow autodiff of cost = -log Z-hat through the tf.while_loop
// of sample_phylogenies (vcsmc.py:445-447, 488-491, 534).  Here it is written out by hand over the node pool
// the forward sweep already keeps (DESIGN.md "VI step"):
//   * discrete choices (resampling indices, pair picks, every gather index) are constants;
//   * branch lengths are reparameterised samples b = -log(U)/lambda (tfp Exponential, vcsmc.py:353-356), so
//     the gradient reaches the rates through them;
//   * the expm, the merges, log, logsumexp are differentiated.
// Accumulation orders are fixed (no floating-point atomics): a gradient is reproducible run to run.
// Parity: oracle/cpu_grad.py (which is checked against central differences), to a relative 1e-9.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phylo_math.h"

#define PG_PART 36                     // per-(node, tile) partial sums: Pl_bar[16], Pr_bar[16], pi_bar[4]
#define PG_NODEG 22                    // per-node results: bl_bar, br_bar, Q_bar[16], pi_bar[4]
#include "phylo_revlists.h"           // PG_PCHUNK, PG_HCHUNK, PG_FREE_PARENT and the host side of the integer lists
#define PG_NT 256                      // sites per workgroup of pg_nodes (4 steps of 64 sites)
#define PG_RED_STRIDE 68                // doubles per value row of the LDS reductions: the 48 reading lanes spread over all banks

#define PG_XCH 16                       // (adopter, slot) entries per chunk of pg_twist_xchunks

// The twisted proposal's part of the graph (PHYLO_TWISTING | PHYLO_KEEP_GRAPH).  Rows: one per (rank event, particle,
// sub-sample j = t M + m of pair t), event r's rows at joff[r] + k J_r + j with J_r = C(N-r, 2) M.
struct pg_twist {
    int M;
    const int64_t* joff;               // [R+1]
    const int32_t* roots_ad;           // [R][K][N] adopted (resampled) root tables
    const double* tw_b;                // [rows][2] branch lengths of every sub-sample
    const double* tw_P;                // [rows][32] their transition matrices
    const double* pot;                 // [rows] look-ahead potentials
    const double* chosen;              // [R][K] chosen sub-sample
    double* tau;                       // [rows] d logZ / d pot
    double* ctw;                       // [R][K][N] coefficient of sum_s log(pi . X) of every slot of the adopted table
    double* twpart;                    // [PG_PART][rows] (value-major: pg_twist_finish reads a value of 256 consecutive rows with one
                                       // coalesced load) Pl_bar, Pr_bar, pi_bar of the row's merge, before the factor tau
    double* twnode;                    // [R][K][PG_NODEG] per particle: d_lam_l, d_lam_r terms, Q_bar, pi_bar of its rows
    double* twslice;                   // [K][slices][PG_NODEG]: partial sums of one rank event when J > 256
    // adjoints of the adopted roots: entries (adopter * N + slot) grouped by node, cut into chunks of PG_XCH
    const int32_t* xent;
    const int32_t *xchunk_node, *xchunk_beg, *xchunk_cnt;
    const int32_t* xchunk_part;        // per chunk: (first partner slot) | (one past the last) << 16 -- small launches slice the partners
    const int32_t *xnode_id, *xnode_chunk0, *xnode_nchunks;
    double* tpart;                     // [chunks of one rank event][S][4]
    const uint32_t* pair_hist;         // [N][N][32] or NULL: sites per code pair of two coded leaves (pk_pair_hist)
};

struct pg_args {
    int N, S, K, R, T, jc;             // T tiles of PG_NT sites per node
    int twist;                         // the sweep used the twisted proposal: tw is set
    pg_twist tw;
    const double* leaves;              // [N][S][4]
    const double* pool;                // [R][K][S][4]
    double* adj;                       // [R][K][S][4]: d logZ / d node
    const double* Pmat;                // [R][K][32]
    const double *bl, *br;             // [R][K]
    const double *logw, *lse;          // [R][K], [R] (lse = logsumexp - log K)
    const double *pi, *Q;              // [4], [16]
    const double *lam_l, *lam_r;       // [R]
    const int32_t* child;              // [R][K][2] node ids
    const int32_t* pos;                // [R][K][N]: slot of the adopted table -> position in the new table, -1 = merged
    const int32_t* roots;              // [R+1][K][N]: plane r+1 = root table after rank event r
    const int32_t *ad_off, *ad_idx;    // [R][K+1], [R][K]: adopters of particle k at rank event r, ascending
    const int32_t *par_off, *par_idx;  // [R K + 1], [<= 2 R K]: parents of internal node x, (node * 2 + side), ascending
    const int32_t* heavy_first;        // [R K]: first chunk (within its rank event's chunk range) of a heavy node, else -1
    const int32_t *chunk_beg, *chunk_cnt;   // per chunk: first entry of par_idx, number of entries (<= PG_HCHUNK)
    double* cpart;                     // [chunks of one rank event][S][4]
    const int32_t* slow_flag;          // [R K]: 0: pg_nodes_free; else pg_nodes_rows: (index in slow_idx) << 3 | (bit 0: has parents, bit 1: look-ahead
                                       // entries, bit 2: adopted, left out by the early pg_nodes_free)
    const int32_t* slow_idx;           // the flagged nodes, grouped by rank event (rows form)
    const int32_t* adp;                // NULL, or the adopted (r * K + k), grouped by rank event: pg_G / pg_coeff run on these alone --
                                       // the early pg_nodes_free has written G = C = omega for everybody else
    const unsigned int* mark;          // [R][K] or NULL: node (r, k) was adopted at rank event r + 1 (the lazy sweep's marks)
    int alpha_om;                      // alpha of a FREE parent is omega itself (the early pg_nodes_free: free = nobody adopted it): the
                                       // gathers then need nothing of the coefficient chain
    int chunks_free_only;              // pg_parent_chunks sums the entries of free parents only (one launch for all rank events);
                                       // pg_nodes_rows adds the flagged parents' entries, which end a heavy node's list
    double* slowpart;                  // [flagged nodes][TS][PG_PART] (rows form; else NULL): their partial sums, TS tiles of 256 sites
    unsigned int* row_done;            // [R K][TS] (pg_nodes_rows_all): == row_epoch when that tile of the node's adjoint row is complete
    unsigned int row_epoch;            // this reverse pass's value of row_done (never 0)
    unsigned int* row_timeout;         // set when a wait of pg_nodes_rows_all gave up
    unsigned int* coeff_done;          // [R] or NULL: == row_epoch when pg_coeff of that rank event is complete (pg_nodes_rows_all beside the chain)
    unsigned int* coeff_ticket;        // [R]: workgroups of that launch that have finished (the last one resets it)
    unsigned long long coeff_mask;     // bit r: rank event r has a pg_coeff launch to wait for
    int TS;
    double *om, *G;                    // [R][K]
    double* C;                         // [R][K][N]: coefficient of sum_s log(pi . X) of every root slot after rank event r
    double* part;                      // [R][K][T][PG_PART]
    double* nodeg;                     // [R][K][PG_NODEG] (pg_node_finish writes the two branch adjoints only; the rest: fin_part)
    double* fin_part;                  // [ceil(R K / 32)][20]: Q_bar[16], pi_bar[4] summed over the 32 nodes of a pg_node_finish workgroup
    double* leafpi;                    // [N][4]: sum_s leaf[s][a] / (pi . leaf[s])
    double* leafterm;                  // [K][4]
    double* terms;                     // [R][K][2]
    double* out;                       // [2 R + 20]: d_lam_l, d_lam_r, d_pi, d_Q
};

// ---- small helpers ------------------------------------------------------------------------------
// The reverse pass has a tolerance, not a bit contract (the build keeps -ffp-contract=off for the forward sweep): its
// inner products use fused multiply-adds explicitly, which halves their instruction count.
__device__ __forceinline__ double pg_dot4(double a0, double b0, double a1, double b1, double a2, double b2, double a3, double b3) {
    return __builtin_fma(a3, b3, __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0)));
}
// 1 / x for a normal x > 0 (a site likelihood): v_rcp_f64 and two Newton steps -- a quarter of the instructions of the IEEE
// division and exact to an ulp or two, which is all a pass with a 1e-9 tolerance needs
__device__ __forceinline__ double pg_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double pg_wave_sum(double v) {          // fixed butterfly: same result on every lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// sum of `v` over the 256 threads of the workgroup, result valid on thread 0 (fixed order)
__device__ __forceinline__ double pg_block_sum(double v, double* sh /*[4]*/) {
    v = pg_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

__device__ __forceinline__ const double* pg_row(const pg_args& a, int id) {
    return id < a.N ? a.leaves + (size_t)id * a.S * 4 : a.pool + (size_t)(id - a.N) * a.S * 4;
}

__device__ __constant__ double pg_inv_k[19] = {0.0, 1.0 / 1, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10,
                                                  1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18};
// Frechet derivative of the matrix exponential, L(A, E) = d/dt exp(A + t E) at t = 0, by a scaled Taylor
// series on the pair (X, dX) (the blocks of exp [[A, E], [0, A]]) and pairwise squaring.  ||A||_1 <= 1/2 after
// scaling and up to 18 terms (chosen from the scaled norm) leave a truncation error below 1e-17.
__device__ inline void pg_expm4_frechet(const double* A0, const double* E0, double* Lout) {
    double A[16], E[16], X[16], D[16], SX[16], SD[16], T1[16], T2[16];
    double norm = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cs = cs + (A0[i * 4 + j] < 0.0 ? -A0[i * 4 + j] : A0[i * 4 + j]);
        if (cs > norm) norm = cs;
    }
    int s = 0;
    double lim = 0.5;
    while (norm > lim && s < 60) { lim = lim * 2.0; ++s; }
    const double sc = pm_from_bits((uint64_t)(1023 - s) << 52);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        A[i] = A0[i] * sc;
        E[i] = E0[i] * sc;
        const double id = (i % 5 == 0) ? 1.0 : 0.0;
        X[i] = id; D[i] = 0.0;                 // term 0
        SX[i] = id; SD[i] = 0.0;
    }
    // terms for a truncation error below 1e-17 of the scaled pair (theta = norm after scaling <= 1/2): theta^n / n! with one term
    // of margin for the derivative series.  Branch lengths are ~Exp(10), so most evaluations need 10, not 18
    const double theta = norm * sc;
    const int nterms = theta <= 0.01 ? 8 : theta <= 0.05 ? 10 : theta <= 0.15 ? 13 : theta <= 0.3 ? 15 : 18;
#pragma unroll 1
    for (int k = 1; k <= nterms; ++k) {
        const double inv = pg_inv_k[k];          // 1 / k, correctly rounded (a division is ~30 dependent instructions per term)
        pm_mm4(X, E, T1);                      // D_k = (X_{k-1} E + D_{k-1} A) / k
        pm_mm4(D, A, T2);
#pragma unroll
        for (int i = 0; i < 16; ++i) D[i] = (T1[i] + T2[i]) * inv;
        pm_mm4(X, A, T1);                      // X_k = X_{k-1} A / k
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            X[i] = T1[i] * inv;
            SX[i] = SX[i] + X[i];
            SD[i] = SD[i] + D[i];
        }
    }
#pragma unroll 1
    for (int q = 0; q < s; ++q) {              // (R, dR) <- (R R, R dR + dR R)
        pm_mm4(SX, SD, T1);
        pm_mm4(SD, SX, T2);
#pragma unroll
        for (int i = 0; i < 16; ++i) SD[i] = T1[i] + T2[i];
        pm_mm4(SX, SX, T1);
#pragma unroll
        for (int i = 0; i < 16; ++i) SX[i] = T1[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Lout[i] = SD[i];
}

// The same series with a matrix spread over the four lanes of a quad, lane i holding row i of X, D and of the sums: row i of a
// product is (row i) x (the right factor), and the right factors of the series are A and E, which every lane holds whole -- the
// loop needs no exchange and is a quarter as long; only the squarings fetch the other rows (quad broadcasts).  Element by element
// the arithmetic of pg_expm4_frechet (the same fused chains in the same order).  All four lanes of a quad must be active.
__device__ __forceinline__ void pg_rowmat(const double (&x)[4], const double* B, double (&y)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double acc = x[0] * B[j];
        acc = pm_fma(x[1], B[4 + j], acc);
        acc = pm_fma(x[2], B[8 + j], acc);
        y[j] = pm_fma(x[3], B[12 + j], acc);
    }
}
template <int I> __device__ __forceinline__ double pg_quad(double v);
__device__ __forceinline__ void pg_quad_gather(const double (&rowv)[4], double* full) {      // the whole matrix from the quad's four rows
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        full[0 * 4 + j] = pg_quad<0>(rowv[j]);
        full[1 * 4 + j] = pg_quad<1>(rowv[j]);
        full[2 * 4 + j] = pg_quad<2>(rowv[j]);
        full[3 * 4 + j] = pg_quad<3>(rowv[j]);
    }
}
__device__ inline void pg_expm4_frechet_row(const double* A0, const double* E0, int myrow, double (&Lrow)[4]) {
    double A[16], E[16];
    double norm = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cs = cs + (A0[i * 4 + j] < 0.0 ? -A0[i * 4 + j] : A0[i * 4 + j]);
        if (cs > norm) norm = cs;
    }
    int s = 0;
    double lim = 0.5;
    while (norm > lim && s < 60) { lim = lim * 2.0; ++s; }
    const double sc = pm_from_bits((uint64_t)(1023 - s) << 52);
#pragma unroll
    for (int i = 0; i < 16; ++i) { A[i] = A0[i] * sc; E[i] = E0[i] * sc; }
    double X[4], D[4], SX[4], SD[4], T1[4], T2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double id = j == myrow ? 1.0 : 0.0;
        X[j] = id; D[j] = 0.0; SX[j] = id; SD[j] = 0.0;
    }
    const double theta = norm * sc;
    const int nterms = theta <= 0.01 ? 8 : theta <= 0.05 ? 10 : theta <= 0.15 ? 13 : theta <= 0.3 ? 15 : 18;
#pragma unroll 1
    for (int k = 1; k <= nterms; ++k) {
        const double inv = pg_inv_k[k];          // 1 / k, correctly rounded (a division is ~30 dependent instructions per term)
        pg_rowmat(X, E, T1);                   // D_k = (X_{k-1} E + D_{k-1} A) / k
        pg_rowmat(D, A, T2);
#pragma unroll
        for (int j = 0; j < 4; ++j) D[j] = (T1[j] + T2[j]) * inv;
        pg_rowmat(X, A, T1);                   // X_k = X_{k-1} A / k
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            X[j] = T1[j] * inv;
            SX[j] = SX[j] + X[j];
            SD[j] = SD[j] + D[j];
        }
    }
#pragma unroll 1
    for (int q = 0; q < s; ++q) {              // (R, dR) <- (R R, R dR + dR R); the quad's lanes share s
        double FX[16], FD[16];
        pg_quad_gather(SX, FX);
        pg_quad_gather(SD, FD);
        pg_rowmat(SX, FD, T1);
        pg_rowmat(SD, FX, T2);
#pragma unroll
        for (int j = 0; j < 4; ++j) SD[j] = T1[j] + T2[j];
        pg_rowmat(SX, FX, T1);
#pragma unroll
        for (int j = 0; j < 4; ++j) SX[j] = T1[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Lrow[j] = SD[j];
}

// ---- copies between device memory and PINNED host memory by a kernel (the host pointer is device-visible): the integer lists of
// the reverse pass go down (ancestors, children) and up (adopters, parents) once per training step, a few hundred KB each, in
// the middle of a chain of dependent launches -- where a copy through the DMA engine costs 30-40 us of start-up, a launch 5.
// Up to PG_COPY_N ranges per launch; 4-byte words, grid-stride.
#define PG_COPY_N 5
struct pg_copy3 {
    const uint32_t* src[PG_COPY_N];
    uint32_t* dst[PG_COPY_N];
    size_t n[PG_COPY_N];                                    // words
};
__global__ __launch_bounds__(256) void pg_copy_words(pg_copy3 a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int q = 0; q < PG_COPY_N; ++q)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[q]; i += stride) a.dst[q][i] = a.src[q][i];
}

// ---- g1: omega = softmax_k(log w_r) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void pg_omega(pg_args a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.R * a.K) return;
    const int r = t / a.K;
    a.om[t] = pm_exp((a.logw[t] - a.lse[r]) - pm_log((double)a.K));
}

// ---- g2: G_r[k] = d logZ / d ll_r[k] = omega_r[k] - sum of omega_{r+1} over the particles that adopt k -----
// one wave per (r, k): a surviving particle can have ~K adopters
__global__ __launch_bounds__(64) void pg_G(pg_args a) {
    const int t = a.adp ? a.adp[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    const int r = t / a.K, k = t - r * a.K;
    double sub = 0.0;
    if (r + 1 < a.R) {
        const int32_t* off = a.ad_off + (size_t)(r + 1) * (a.K + 1);
        const int32_t* idx = a.ad_idx + (size_t)(r + 1) * a.K;
        const int beg = off[k], end = off[k + 1];
        if (end > beg) {
            for (int j0 = beg + lane; j0 < end; j0 += 256) {       // four adopters per lane in flight (index -> weight)
                int ix[4];
                double w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ix[u] = idx[j0 + 64 * u < end ? j0 + 64 * u : end - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = a.om[(size_t)(r + 1) * a.K + ix[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + 64 * u < end) sub = sub + w[u];
            }
            sub = pg_wave_sum(sub);
        }
    }
    if (lane == 0) a.G[t] = a.om[t] - sub;
    // (the tickets of the coefficient launches behind this one start at zero whatever an earlier, failed pass left in them)
    if (a.coeff_ticket && blockIdx.x == 0 && lane < a.R) a.coeff_ticket[lane] = 0u;
}

// ---- g3: root-slot coefficients, one rank event per launch (newest first) ---------------------------------
// C_r[k][slot] = G_r[k] + sum over adopters k' of C_{r+1}[k'][position of that slot in k''s new table]
// grid (K -- or the adopted particles of rank event r, adp[adp0 ...] --, slot groups); 4 waves per workgroup, one slot each.
// Values handed from one workgroup to another INSIDE a launch (pg_coeff_all, pg_nodes_rows_all) are stored and loaded at agent
// scope (device-coherent accesses that pass the per-XCD L2), ordered by s_waitcnt + the workgroup barrier around a relaxed
// completion word: no cache-wide write-back / invalidate per hand-off (those cost every kernel on the GPU its L2 contents -- the
// parents' sort beside the chains ran at half speed -- and made a hand-off as slow as a launch boundary).
__device__ __forceinline__ double pg_ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pg_st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bx, by: the workgroup's place in rank event r's grid (adopted particle, group of four slots); total: that grid's workgroups.
// wait_word != NULL (pg_coeff_all): rank event r + 1's coefficients are complete when *wait_word == wait_for.  Everything that
// does not depend on them -- the adopters' indices and positions of up to PG_COEFF_PRE x 256 adopters, two dependent loads each --
// is loaded BEFORE the wait; behind it there is one round of coefficient loads, the sum, the store and the completion word.
#define PG_COEFF_PRE 4
__device__ __forceinline__ void pg_coeff_body(const pg_args& a, int r, int adp0, int bx, int by, unsigned int total,
                                              const unsigned int* wait_word, unsigned int wait_for) {
    const int k = a.adp ? a.adp[adp0 + bx] - r * a.K : bx, lane = threadIdx.x & 63;
    const int slot = by * 4 + (threadIdx.x >> 6);
    const int n1 = a.N - r - 1;
    const bool active = slot < n1;                           // (per wave)
    double g = 0.0;
    int off = 0, cnt = 0;
    size_t row[PG_COEFF_PRE][4];
    int p[PG_COEFF_PRE][4];
    double tv[PG_COEFF_PRE][4];
    const size_t base = (size_t)(r + 1) * a.K;
    if (active) {
        g = a.G[(size_t)r * a.K + k];
        if (r + 1 < a.R) {
            const int32_t* o = a.ad_off + (size_t)(r + 1) * (a.K + 1);
            off = o[k];
            cnt = o[k + 1] - off;
        }
        if (cnt > 0) {
            // three dependent loads per adopter (index -> position -> coefficient): four adopters per lane and iteration in flight, each
            // level's loads issued together (clamped index: no branch around a load); added in the same order as one at a time
            const int32_t* idx = a.ad_idx + (size_t)(r + 1) * a.K + off;
#pragma unroll
            for (int it = 0; it < PG_COEFF_PRE; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = lane + 256 * it + 64 * u;
                    row[it][u] = (base + idx[jj < cnt ? jj : cnt - 1]) * a.N;
                }
#pragma unroll
            for (int it = 0; it < PG_COEFF_PRE; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p[it][u] = a.pos[row[it][u] + slot];
                    tv[it][u] = a.twist ? a.tw.ctw[row[it][u] + slot] : 0.0;   // the adopter's potentials subtract post() of every adopted root
                }
        }
    }
    if (wait_word) {                                         // (uniform)
        if (threadIdx.x == 0) {
            unsigned int spins = 0;
            while (__hip_atomic_load(wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_for) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 18)) {
                    __hip_atomic_store(a.row_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (active) {
        double v = 0.0;
        if (cnt > 0) {
            double cv[PG_COEFF_PRE][4];
#pragma unroll
            for (int it = 0; it < PG_COEFF_PRE; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* cp = a.C + row[it][u] + (p[it][u] >= 0 ? p[it][u] : 0);
                    cv[it][u] = a.coeff_done ? pg_ld_agent(cp) : *cp;
                }
#pragma unroll
            for (int it = 0; it < PG_COEFF_PRE; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lane + 256 * it + 64 * u < cnt) {
                        if (p[it][u] >= 0) v = v + cv[it][u];
                        if (a.twist) v = v + tv[it][u];
                    }
            const int32_t* idx = a.ad_idx + (size_t)(r + 1) * a.K + off;
            for (int j0 = lane + 256 * PG_COEFF_PRE; j0 < cnt; j0 += 256) {      // (more than 1024 adopters of one particle)
                size_t rw[4];
                int pp[4];
                double cw[4], tw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 64 * u < cnt ? j0 + 64 * u : cnt - 1;
                    rw[u] = (base + idx[j]) * a.N;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pp[u] = a.pos[rw[u] + slot];
                    tw[u] = a.twist ? a.tw.ctw[rw[u] + slot] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* cp = a.C + rw[u] + (pp[u] >= 0 ? pp[u] : 0);
                    cw[u] = a.coeff_done ? pg_ld_agent(cp) : *cp;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + 64 * u < cnt) {
                        if (pp[u] >= 0) v = v + cw[u];
                        if (a.twist) v = v + tw[u];
                    }
            }
            v = pg_wave_sum(v);
        }
        if (lane == 0) {
            double* Ck = a.C + ((size_t)r * a.K + k) * a.N;
            if (a.coeff_done) pg_st_agent(Ck + slot, g + v); else Ck[slot] = g + v;
        }
    }
    if (a.coeff_done) {                                      // (uniform) somebody waits for this launch inside another one: the last
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // workgroup to finish says that rank event r's coefficients are complete
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int t = __hip_atomic_fetch_add(a.coeff_ticket + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (pg_coeff_all's workgroups wait for the ticket itself to reach the rank event's count -- one hop less than the word, which
            //  is for pg_nodes_rows_all; pg_G zeroes the tickets at the head of every pass)
            if (t == total - 1) __hip_atomic_store(a.coeff_done + r, a.row_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ __launch_bounds__(256) void pg_coeff(pg_args a, int r, int adp0) {
    pg_coeff_body(a, r, adp0, (int)blockIdx.x, (int)blockIdx.y, gridDim.x * gridDim.y, nullptr, 0u);
}
// The whole coefficient chain as ONE launch: the grids of all rank events one after the other, newest rank event first (a workgroup
// is dispatched after everything it waits for); a workgroup of rank event r waits for the completion word of rank event r + 1
// (coeff_done, set by the last workgroup of that rank event to finish), bounded like the waits of pg_nodes_rows_all.  Ten dependent
// launches of 11 us + 6 us between them become one launch with a short hand-off per rank event.
struct pg_coeff_plan { int first[66]; int adp0[66]; int ny[66]; };   // by rank event: first workgroup (descending r), adp offset, slot groups
__global__ __launch_bounds__(256) void pg_coeff_all(pg_args a, pg_coeff_plan pl) {
    const int b = (int)blockIdx.x;
    int r = a.R - 2;
    while (r > 0 && b >= pl.first[r - 1]) --r;               // first[] grows as r falls: first[R - 2] = 0
    const int local = b - pl.first[r], ny = pl.ny[r];
    const int bx = local / ny, by = local - bx * ny;
    const int total = r > 0 ? pl.first[r - 1] - pl.first[r] : (int)gridDim.x - pl.first[r];
    const bool waits = r + 1 < a.R - 1 && ((a.coeff_mask >> (r + 1)) & 1ull);   // rank event r + 1 has coefficients of its own
    const unsigned int newer = waits ? (unsigned int)(pl.first[r] - pl.first[r + 1]) : 0u;   // that rank event's workgroups
    pg_coeff_body(a, r, pl.adp0[r], bx, by, (unsigned int)total, waits ? a.coeff_ticket + r + 1 : nullptr, newer);
}

// ---- g4: per-leaf sums for d/d pi of the leaf terms -----------------------------------------------------
__global__ __launch_bounds__(256) void pg_leafpi(pg_args a) {
    __shared__ double sh[4];
    const int x = blockIdx.x;
    const double* row = a.leaves + (size_t)x * a.S * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s = threadIdx.x; s < a.S; s += 256) {
        const double l0 = row[s * 4], l1 = row[s * 4 + 1], l2 = row[s * 4 + 2], l3 = row[s * 4 + 3];
        const double lik = ((a.pi[0] * l0 + a.pi[1] * l1) + a.pi[2] * l2) + a.pi[3] * l3;
        const double inv = 1.0 / lik;
        acc[0] = acc[0] + l0 * inv; acc[1] = acc[1] + l1 * inv; acc[2] = acc[2] + l2 * inv; acc[3] = acc[3] + l3 * inv;
    }
    for (int q = 0; q < 4; ++q) {
        const double t = pg_block_sum(acc[q], sh);
        if (threadIdx.x == 0) a.leafpi[x * 4 + q] = t;
    }
}

// leaves that still are roots after rank event 0 enter ll through pi . leaf[s]
__global__ __launch_bounds__(256) void pg_leafterm(pg_args a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.K) return;
    const int32_t* tab = a.roots + ((size_t)1 * a.K + k) * a.N;
    const double* Ck = a.C + (size_t)k * a.N;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = 0; p < a.N - 1; ++p) {
        const int x = tab[p];
        if (x < a.N) {
            const double c = Ck[p];
            for (int q = 0; q < 4; ++q) acc[q] = acc[q] + c * a.leafpi[x * 4 + q];
        }
    }
    if (a.twist)                                            // rank event 0 adopts the leaves themselves
        for (int x = 0; x < a.N; ++x) {
            const double c = a.tw.ctw[(size_t)k * a.N + x];
            for (int q = 0; q < 4; ++q) acc[q] = acc[q] + c * a.leafpi[x * 4 + q];
        }
    for (int q = 0; q < 4; ++q) a.leafterm[k * 4 + q] = acc[q];
}

// ---- g5: node adjoints of one rank event (newest first) ---------------------------------------------------
// Xbar = alpha pi / (pi . X) + sum over parents (Xbar_parent o (sib P_sib)) P_me^T.
// Quad form: lane 4 q + j owns state j of site q, so every load and store is 8 B per lane, 512 B contiguous per
// wave; the other three states of the site come from DPP quad broadcasts, and lane j accumulates column j of
// Pl_bar / Pr_bar (9 running sums per lane instead of 36).
// After the first resamplings only a handful of lineages survive, so a few nodes have ~K parents and the rest
// none: a node with more than PG_PCHUNK parents has its parent list cut into chunks of PG_HCHUNK that
// pg_parent_chunks sums in parallel; pg_nodes then adds the chunk sums in order.  A light node gathers inline.
template <int I>
__device__ __forceinline__ double pg_quad(double v) {       // the value held by lane I of my quad
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, I * 0x55, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, I * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int X>
__device__ __forceinline__ double pg_quad_sum_step(double v) {   // the value of lane (me ^ X) of my quad, X = 1 or 2
    int lo = __double2loint(v), hi = __double2hiint(v);
    constexpr int ctrl = X == 1 ? 0xB1 : 0x4E;                  // quad_perm [1,0,3,2] / [2,3,0,1]
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// (a par_idx entry carries PG_FREE_PARENT when the parent's adjoint row is not stored: the gather recomputes it from alpha_parent,
// staged in shA)
__device__ __forceinline__ double pg_alpha_of(const pg_args& a, int pn) {
    if (a.alpha_om) return a.om[pn];                        // (read for free parents only; a flagged parent's row is stored)
    const int rp = pn / a.K;
    return a.C[(size_t)pn * a.N + (a.N - rp - 2)];
}
__device__ __forceinline__ void pg_stage_parents(const pg_args& a, int c0, int nc, double (*shP)[32], int* shE, int* shSib, double* shA) {
    const int e = threadIdx.x >> 5, q = threadIdx.x & 31;
    if (e < nc) {
        const int enc = a.par_idx[c0 + e];
        const int pn = (enc & (PG_FREE_PARENT - 1)) >> 1, side = enc & 1;
        shP[e][q] = a.Pmat[(size_t)pn * 32 + q];
        if (q == 0) { shE[e] = enc; shSib[e] = a.child[(size_t)pn * 2 + (1 - side)]; }
        if (q == 1) shA[e] = pg_alpha_of(a, pn);
    }
}

// xb (state j of one site) += contributions of the nc staged parents; soff = s * 4 + j.  All 2 nc loads are issued before the
// first is used (index clamped, so no branch surrounds a load): a loop with a load inside pays one memory latency per parent, and
// that -- not bandwidth or arithmetic -- was what pg_parent_chunks and the heavy nodes of pg_nodes cost.
__device__ __forceinline__ double pg_parent_quad(const pg_args& a, size_t soff, int j, int nc, const double (*shP)[32],
                                                 const int* shE, const int* shSib, const double* shA, double me, double xb) {
    const size_t row = (size_t)a.S * 4;
    double xpv[PG_PCHUNK], sbv[PG_PCHUNK];
#pragma unroll
    for (int e = 0; e < PG_PCHUNK; ++e) {
        const int ee = e < nc ? e : nc - 1;
        const int enc = __builtin_amdgcn_readfirstlane(shE[ee]);
        xpv[e] = 0.0;
        if (!(enc & PG_FREE_PARENT)) xpv[e] = a.adj[(size_t)(enc >> 1) * row + soff];
        sbv[e] = pg_row(a, shSib[ee])[soff];
    }
    const double m0 = pg_quad<0>(me), m1 = pg_quad<1>(me), m2 = pg_quad<2>(me), m3 = pg_quad<3>(me);
    const double pj = a.pi[j];
#pragma unroll
    for (int e = 0; e < PG_PCHUNK; ++e) {
        if (e < nc) {                                        // wave-uniform
            const int enc = __builtin_amdgcn_readfirstlane(shE[e]);
            const int side = enc & 1;
            const double* Psib = shP[e] + (1 - side) * 16;
            const double* Pme = shP[e] + side * 16;
            const double sb = sbv[e];
            const double b0 = pg_quad<0>(sb), b1 = pg_quad<1>(sb), b2 = pg_quad<2>(sb), b3 = pg_quad<3>(sb);
            const double w = pg_dot4(b0, Psib[j], b1, Psib[4 + j], b2, Psib[8 + j], b3, Psib[12 + j]);
            double xp = xpv[e];
            if (enc & PG_FREE_PARENT) {                      // alpha_p pi / (pi . X_p),  X_p = (me P_me) o (sib P_sib)
                const double u = pg_dot4(m0, Pme[j], m1, Pme[4 + j], m2, Pme[8 + j], m3, Pme[12 + j]);
                double lik = pj * (u * w);
                lik = lik + pg_quad_sum_step<1>(lik);
                lik = lik + pg_quad_sum_step<2>(lik);
                xp = (shA[e] * pj) * pg_rcp(lik);
            }
            const double t = xp * w;
            const double t0 = pg_quad<0>(t), t1 = pg_quad<1>(t), t2 = pg_quad<2>(t), t3 = pg_quad<3>(t);
            xb = xb + pg_dot4(t0, Pme[j * 4], t1, Pme[j * 4 + 1], t2, Pme[j * 4 + 2], t3, Pme[j * 4 + 3]);
        }
    }
    return xb;
}

// grid (groups of 16 PG_CSTEPS sites, chunks -- of rank event chunk0's range, or with chunks_free_only of ALL rank events, chunk0
// = the first of the launch): cpart[chunk][s] = sum of the chunk's parent contributions (free parents' only with chunks_free_only).  The chunk's
// (up to PG_HCHUNK = 4 PG_PCHUNK) parents are staged at once and each of the four waves gathers its own PG_PCHUNK of them for the
// same 16 PG_CSTEPS sites (all sibling rows of the steps in flight together); the four partial sums are added in wave order.
// (One wave after the other over 64 sites -- the first version -- paid a staging and a gather latency per PG_PCHUNK parents:
// 19.5 us per launch.  A workgroup's time is staging, ~3 us, + ~2 us of arithmetic per step: one step needs two passes over the
// chip on an average rank event, four steps make the pass too long: 18.7 / 21.7 us.)
#define PG_CSTEPS 2
__global__ __launch_bounds__(256) void pg_parent_chunks(pg_args a, int chunk0) {
    static_assert(PG_HCHUNK == 4 * PG_PCHUNK, "one wave per PG_PCHUNK parents of a chunk");
    __shared__ double shP[PG_HCHUNK][32];
    __shared__ int shE[PG_HCHUNK];
    __shared__ int shSib[PG_HCHUNK];
    __shared__ double shA[PG_HCHUNK];
    __shared__ int shMe;
    __shared__ double shX[4][PG_CSTEPS][64];
    const int ci = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = a.chunk_beg[chunk0 + ci], cnt = a.chunk_cnt[chunk0 + ci];
    const int j = lane & 3;
    const size_t row = (size_t)a.S * 4;
    size_t soff[PG_CSTEPS];
    bool live[PG_CSTEPS];
#pragma unroll
    for (int it = 0; it < PG_CSTEPS; ++it) {
        const int s = blockIdx.x * (16 * PG_CSTEPS) + it * 16 + (lane >> 2);
        live[it] = s < a.S;
        soff[it] = (size_t)(live[it] ? s : a.S - 1) * 4 + j;
    }
    for (int i = tid; i < cnt * 32; i += 256) {
        const int e = i >> 5, q = i & 31;
        const int enc = a.par_idx[c0 + e];
        const int pn = (enc & (PG_FREE_PARENT - 1)) >> 1, side = enc & 1;
        shP[e][q] = a.Pmat[(size_t)pn * 32 + q];
        if (q == 0) { shE[e] = enc; shSib[e] = a.child[(size_t)pn * 2 + (1 - side)]; }
        if (q == 1) shA[e] = pg_alpha_of(a, pn);
        if (i == 2) shMe = a.child[(size_t)pn * 2 + side];   // the node all these are parents of
    }
    __syncthreads();
    const int nc = cnt - wv * PG_PCHUNK < PG_PCHUNK ? cnt - wv * PG_PCHUNK : PG_PCHUNK;
    double xb[PG_CSTEPS];
#pragma unroll
    for (int it = 0; it < PG_CSTEPS; ++it) xb[it] = 0.0;
    if (nc > 0) {
        const int e0 = wv * PG_PCHUNK;
        const double* merow = pg_row(a, shMe);
        double me[PG_CSTEPS], sbv[PG_CSTEPS][PG_PCHUNK];
#pragma unroll
        for (int it = 0; it < PG_CSTEPS; ++it) me[it] = merow[soff[it]];
#pragma unroll
        for (int e = 0; e < PG_PCHUNK; ++e) {
            const double* sr = pg_row(a, shSib[e0 + (e < nc ? e : nc - 1)]);
#pragma unroll
            for (int it = 0; it < PG_CSTEPS; ++it) sbv[it][e] = sr[soff[it]];
        }
        const double pj = a.pi[j];
#pragma unroll
        for (int e = 0; e < PG_PCHUNK; ++e) {
            if (e < nc) {                                    // wave-uniform
                const int enc = __builtin_amdgcn_readfirstlane(shE[e0 + e]);
                if (a.chunks_free_only && !(enc & PG_FREE_PARENT)) continue;   // a flagged parent: pg_nodes_rows adds it
                const int side = enc & 1;
                const double* Psib = shP[e0 + e] + (1 - side) * 16;
                const double* Pme = shP[e0 + e] + side * 16;
                const double al = shA[e0 + e];
#pragma unroll
                for (int it = 0; it < PG_CSTEPS; ++it) {
                    const double sb = sbv[it][e];
                    const double b0 = pg_quad<0>(sb), b1 = pg_quad<1>(sb), b2 = pg_quad<2>(sb), b3 = pg_quad<3>(sb);
                    const double w = pg_dot4(b0, Psib[j], b1, Psib[4 + j], b2, Psib[8 + j], b3, Psib[12 + j]);
                    double xp;
                    if (enc & PG_FREE_PARENT) {              // alpha_p pi / (pi . X_p),  X_p = (me P_me) o (sib P_sib)
                        const double m = me[it];
                        const double m0 = pg_quad<0>(m), m1 = pg_quad<1>(m), m2 = pg_quad<2>(m), m3 = pg_quad<3>(m);
                        const double u = pg_dot4(m0, Pme[j], m1, Pme[4 + j], m2, Pme[8 + j], m3, Pme[12 + j]);
                        double lik = pj * (u * w);
                        lik = lik + pg_quad_sum_step<1>(lik);
                        lik = lik + pg_quad_sum_step<2>(lik);
                        xp = (al * pj) * pg_rcp(lik);
                    } else {                                 // a parent with parents of its own (rare in a heavy node's list): its stored row
                        xp = a.adj[(size_t)(enc >> 1) * row + soff[it]];
                    }
                    const double t = xp * w;
                    const double t0 = pg_quad<0>(t), t1 = pg_quad<1>(t), t2 = pg_quad<2>(t), t3 = pg_quad<3>(t);
                    xb[it] = xb[it] + pg_dot4(t0, Pme[j * 4], t1, Pme[j * 4 + 1], t2, Pme[j * 4 + 2], t3, Pme[j * 4 + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < PG_CSTEPS; ++it) shX[wv][it][lane] = xb[it];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int it = 0; it < PG_CSTEPS; ++it)
            if (live[it]) a.cpart[(size_t)ci * row + soff[it]] = ((shX[0][it][lane] + shX[1][it][lane]) + shX[2][it][lane]) + shX[3][it][lane];
    }
}

// grid (tiles of PG_NT sites, K)
__global__ __launch_bounds__(256) void pg_nodes(pg_args a, int r) {
    __shared__ double shP[PG_PCHUNK][32];
    __shared__ int shE[PG_PCHUNK];
    __shared__ double shA[PG_PCHUNK];
    __shared__ int shSib[PG_PCHUNK];
    __shared__ double shOwn[32];
    __shared__ double shR[4][9][4];
    const int k = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int j = tid & 3, q = tid >> 2;
    const size_t node = (size_t)r * a.K + k;
    const size_t row = (size_t)a.S * 4;
    const double alpha = a.C[node * a.N + (a.N - r - 2)];
    const double p0 = a.pi[0], p1 = a.pi[1], p2 = a.pi[2], p3 = a.pi[3];
    const double pj = a.pi[j];
    const int pbeg = a.par_off[node], pend = a.par_off[node + 1];
    const int hv = a.heavy_first[node];
    const int np = pend - pbeg;
    const int nch = hv >= 0 ? (np + PG_HCHUNK - 1) / PG_HCHUNK : 0;
    const bool has_x = (a.slow_flag[node] & 2) != 0;
    if (tid < 32) shOwn[tid] = a.Pmat[node * 32 + tid];
    if (hv < 0 && np > 0) pg_stage_parents(a, pbeg, np, shP, shE, shSib, shA);
    __syncthreads();
    const double* Lrow = pg_row(a, a.child[node * 2]);
    const double* Rrow = pg_row(a, a.child[node * 2 + 1]);
    const double* xrow = a.pool + node * row;
    double* orow = a.adj + node * row;
    const double* Pl = shOwn;
    const double* Pr = shOwn + 16;
    double acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.0;
    const int s_end = (tile + 1) * PG_NT < a.S ? (tile + 1) * PG_NT : a.S;
    // the tile is PG_NT / 64 = 4 steps of 64 sites: all of a thread's loads are issued before the first is used
    double xs[PG_NT / 64], Ls[PG_NT / 64], Rs[PG_NT / 64];
#pragma unroll
    for (int it = 0; it < PG_NT / 64; ++it) {
        const int s = tile * PG_NT + q + it * 64;
        xs[it] = Ls[it] = Rs[it] = 0.0;
        if (s < s_end) {
            const size_t soff = (size_t)s * 4 + j;
            xs[it] = xrow[soff];
            Ls[it] = Lrow[soff];
            Rs[it] = Rrow[soff];
        }
    }
#pragma unroll
    for (int it = 0; it < PG_NT / 64; ++it) {
        const int s = tile * PG_NT + q + it * 64;
        if (s >= s_end) continue;
        const size_t soff = (size_t)s * 4 + j;
        const double x = xs[it];
        const double Lj = Ls[it], Rj = Rs[it];
        const double x0 = pg_quad<0>(x), x1 = pg_quad<1>(x), x2 = pg_quad<2>(x), x3 = pg_quad<3>(x);
        const double lik = pg_dot4(p0, x0, p1, x1, p2, x2, p3, x3);
        const double inv = alpha / lik;
        double xb = pj * inv;
        if (has_x) xb = xb + orow[soff];                    // what the look-ahead merges of later rank events left (pg_twist_xsum)
        acc[8] = __builtin_fma(x, inv, acc[8]);
        if (hv >= 0) {
            const double* cp = a.cpart + (size_t)hv * row + soff;
            int c = 0;
            for (; c + 4 <= nch; c += 4) {
                const double q0 = cp[(size_t)c * row], q1 = cp[(size_t)(c + 1) * row], q2 = cp[(size_t)(c + 2) * row],
                             q3 = cp[(size_t)(c + 3) * row];
                xb = (((xb + q0) + q1) + q2) + q3;
            }
            for (; c < nch; ++c) xb = xb + cp[(size_t)c * row];
        } else if (np > 0) {
            xb = pg_parent_quad(a, soff, j, np, shP, shE, shSib, shA, x, xb);
        }
        orow[soff] = xb;
        const double L0 = pg_quad<0>(Lj), L1 = pg_quad<1>(Lj), L2 = pg_quad<2>(Lj), L3 = pg_quad<3>(Lj);
        const double R0 = pg_quad<0>(Rj), R1 = pg_quad<1>(Rj), R2 = pg_quad<2>(Rj), R3 = pg_quad<3>(Rj);
        const double u = pg_dot4(L0, Pl[j], L1, Pl[4 + j], L2, Pl[8 + j], L3, Pl[12 + j]);
        const double v = pg_dot4(R0, Pr[j], R1, Pr[4 + j], R2, Pr[8 + j], R3, Pr[12 + j]);
        const double tl = xb * v, tr = xb * u;
        acc[0] = __builtin_fma(L0, tl, acc[0]); acc[1] = __builtin_fma(L1, tl, acc[1]);
        acc[2] = __builtin_fma(L2, tl, acc[2]); acc[3] = __builtin_fma(L3, tl, acc[3]);
        acc[4] = __builtin_fma(R0, tr, acc[4]); acc[5] = __builtin_fma(R1, tr, acc[5]);
        acc[6] = __builtin_fma(R2, tr, acc[6]); acc[7] = __builtin_fma(R3, tr, acc[7]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {                           // over the 16 quads of the wave, state j stays in lane j
        double v = acc[i];
        v = v + __shfl_xor(v, 4, 64);
        v = v + __shfl_xor(v, 8, 64);
        v = v + __shfl_xor(v, 16, 64);
        v = v + __shfl_xor(v, 32, 64);
        if ((tid & 63) < 4) shR[tid >> 6][i][j] = v;
    }
    __syncthreads();
    if (tid < PG_PART) {                                    // part[i * 4 + j] = Pl_bar[i][j], 16 + .. = Pr_bar, 32 + j = pi_bar[j]
        const int i = tid < 32 ? (tid >> 2) : 8, jj = tid & 3;
        a.part[(node * a.T + tile) * PG_PART + tid] = ((shR[0][i][jj] + shR[1][i][jj]) + shR[2][i][jj]) + shR[3][i][jj];
    }
}

// pg_parent_chunks in ROW form: grid (groups of 64 sites, chunks), lane = site (all four states: whole 32-byte rows per load, no
// quad broadcasts -- the quad form above spends 24 of its ~50 instructions per quarter site on DPP moves), wave w gathers
// parents w PG_PCHUNK .. of the chunk for the same 64 sites, the four partial sums are added in wave order.  Every sibling row
// of a wave's parents is loaded before the first is used.  Same sum per chunk up to the order inside a 4-term dot product.
__global__ __launch_bounds__(256) void pg_parent_chunks_rows(pg_args a, int chunk0) {
    static_assert(PG_HCHUNK == 4 * PG_PCHUNK, "one wave per PG_PCHUNK parents of a chunk");
    __shared__ double shP[PG_HCHUNK][32];
    __shared__ int shE[PG_HCHUNK];
    __shared__ int shSib[PG_HCHUNK];
    __shared__ double shA[PG_HCHUNK];
    __shared__ int shMe;
    __shared__ double shX[4][4][64];
    const int ci = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = a.chunk_beg[chunk0 + ci], cnt = a.chunk_cnt[chunk0 + ci];
    const size_t row = (size_t)a.S * 4;
    const int s = blockIdx.x * 64 + lane;
    const bool live = s < a.S;
    const size_t so = (size_t)(live ? s : a.S - 1) * 4;
    for (int i = tid; i < cnt * 32; i += 256) {
        const int e = i >> 5, q = i & 31;
        const int enc = a.par_idx[c0 + e];
        const int pn = (enc & (PG_FREE_PARENT - 1)) >> 1, side = enc & 1;
        shP[e][q] = a.Pmat[(size_t)pn * 32 + q];
        if (q == 0) { shE[e] = enc; shSib[e] = a.child[(size_t)pn * 2 + (1 - side)]; }
        if (q == 1) shA[e] = pg_alpha_of(a, pn);
        if (i == 2) shMe = a.child[(size_t)pn * 2 + side];   // the node all these are parents of
    }
    __syncthreads();
    const int nc = cnt - wv * PG_PCHUNK < PG_PCHUNK ? cnt - wv * PG_PCHUNK : PG_PCHUNK;
    double xb[4] = {0.0, 0.0, 0.0, 0.0};
    if (nc > 0) {
        const int e0 = wv * PG_PCHUNK;
        const double* merow = pg_row(a, shMe) + so;
        double x[4], sb[PG_PCHUNK][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = merow[j];
#pragma unroll
        for (int e = 0; e < PG_PCHUNK; ++e) {
            const double* sr = pg_row(a, shSib[e0 + (e < nc ? e : nc - 1)]) + so;
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[e][j] = sr[j];
        }
        const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
#pragma unroll
        for (int e = 0; e < PG_PCHUNK; ++e) {
            if (e < nc) {                                    // wave-uniform
                const int enc = __builtin_amdgcn_readfirstlane(shE[e0 + e]);
                if (a.chunks_free_only && !(enc & PG_FREE_PARENT)) continue;   // a flagged parent: pg_nodes_rows adds it
                const int side = enc & 1;
                const double* Psib = shP[e0 + e] + (1 - side) * 16;
                const double* Pme = shP[e0 + e] + side * 16;
                double w[4], xp[4], t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = pg_dot4(sb[e][0], Psib[j], sb[e][1], Psib[4 + j], sb[e][2], Psib[8 + j], sb[e][3], Psib[12 + j]);
                if (enc & PG_FREE_PARENT) {                  // alpha_p pi / (pi . X_p),  X_p = (me P_me) o (sib P_sib)
                    double lik = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        lik = __builtin_fma(pi[j] * w[j], pg_dot4(x[0], Pme[j], x[1], Pme[4 + j], x[2], Pme[8 + j], x[3], Pme[12 + j]), lik);
                    const double ai = shA[e0 + e] * pg_rcp(lik);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xp[j] = pi[j] * ai;
                } else {                                     // a parent with parents of its own (rare in a heavy node's list): its stored row
                    const double* xpp = a.adj + (size_t)(enc >> 1) * row + so;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xp[j] = xpp[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = xp[j] * w[j];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xb[j] = xb[j] + pg_dot4(t[0], Pme[j * 4], t[1], Pme[j * 4 + 1], t[2], Pme[j * 4 + 2], t[3], Pme[j * 4 + 3]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) shX[wv][j][lane] = xb[j];
    __syncthreads();
    if (wv == 0 && live) {
        double* out = a.cpart + (size_t)ci * row + so;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = ((shX[0][j][lane] + shX[1][j][lane]) + shX[2][j][lane]) + shX[3][j][lane];
    }
}

// The 36 sums of a node over the 256 threads of its workgroup (fixed order): per wave through its own LDS rows, then the four
// waves in order; thread t < PG_PART writes sum t.
struct pg_rows_lds {
    double red[4][12][PG_RED_STRIDE];
    double shW[4][PG_PART];
};
__device__ __forceinline__ void pg_rows_reduce(const double (&acc)[PG_PART], pg_rows_lds& sh, double* out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 2, part = lane & 3;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int i = 0; i < 12; ++i) sh.red[wv][i][lane] = acc[b * 12 + i];
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
        if (q < 12) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v = v + sh.red[wv][q][part + 4 * i];
        }
        v = v + pg_quad_sum_step<1>(v);
        v = v + pg_quad_sum_step<2>(v);
        if (q < 12 && part == 0) sh.shW[wv][b * 12 + q] = v;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < PG_PART) out[tid] = ((sh.shW[0][tid] + sh.shW[1][tid]) + sh.shW[2][tid]) + sh.shW[3][tid];
}

// G = C = omega for every (r, k) nobody adopted (see pg_nodes_free, phase 0): as a launch of its own, thread per (r, k), when the
// long pg_nodes_free runs in the background on another stream and the coefficient chain must not wait for it (phase 3 there).
__global__ __launch_bounds__(256) void pg_fill_free(pg_args a) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.R * a.K || a.mark[t]) return;
    const int r = (int)(t / (size_t)a.K);
    const double om = a.om[t];
    a.G[t] = om;
    for (int slot = 0; slot < a.N - r - 1; ++slot) a.C[t * a.N + slot] = om;
}

// Nodes nobody merged again -- all but the few adopted ones, of ALL rank events in one launch: one WAVE per node (grid R K / 4;
// a flagged node's wave leaves at once: pg_nodes_rows takes it).  Their adjoint is the own term alone, alpha pi / (pi . X), so one
// pass over the sites does everything.  The node's own row is recomputed from the children, (L P_l) o (R P_r) as in the forward
// merge: the stored row is not read, so the sweep need not have stored it (lazy nodes), and the adjoint row is not written
// either -- a child that gathers from such a parent recomputes the own term (PG_FREE_PARENT), which costs less than the HBM
// round trip of the row.  Children are leaves or adopted nodes: a few rows, read by thousands of nodes (L2).  A wave per node,
// not a workgroup: the chain node -> children -> rows -> 36 sums is latency, and sixteen independent chains per CU hide it
// where four workgroups did not (132 us per launch at K = 2048, N = 12); the rows of the next 64 sites are loaded while the
// current ones are used.
// phase 0 (before the host has built any list; needs the marks of a lazy sweep): the nodes nobody adopted -- no adopters, so
//   alpha = G = omega exactly, and no parents; it also writes G = C = omega for them.  The host then flags every adopted node for
//   pg_nodes_rows.  (A marked node that was not adopted -- phylo_sweep_node marks what it writes -- would be lost: after that
//   call the marks do not count and phase 1 is used.)
// phase 3: the same nodes, without the G / C writes (pg_fill_free has made them): the background launch.
// phase 1 (after the lists; a sweep without usable marks): every node without a flag.
__global__ __launch_bounds__(256, 3) void pg_nodes_free(pg_args a, int phase) {
    __shared__ double red[4][12][PG_RED_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t node = (size_t)blockIdx.x * 4 + wv;
    if (node >= (size_t)a.R * a.K) return;
    const bool by_marks = phase == 0 || phase == 3;
    if (by_marks ? a.mark[node] != 0u : a.slow_flag[node] != 0) return;
    const int r = (int)(node / (size_t)a.K);
    const double alpha = by_marks ? a.om[node] : a.C[node * a.N + (a.N - r - 2)];

    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    const double* Pu = a.Pmat + node * 32;
    double Pl[16], Pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Pl[i] = Pu[i]; Pr[i] = Pu[16 + i]; }
    const int cl = a.child[node * 2], cr = a.child[node * 2 + 1];
    const double* Lrow = pg_row(a, cl);
    const double* Rrow = pg_row(a, cr);
    double acc[PG_PART];
#pragma unroll
    for (int i = 0; i < PG_PART; ++i) acc[i] = 0.0;
    double Ln[4], Rn[4];
    {
        const size_t so = (size_t)(lane < a.S ? lane : a.S - 1) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ln[i] = Lrow[so + i]; Rn[i] = Rrow[so + i]; }
    }
    #pragma unroll 1
    for (int s = lane; s < a.S; s += 64) {
        double L[4], Rv[4], u[4], v[4], xb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { L[i] = Ln[i]; Rv[i] = Rn[i]; }
        {
            const size_t sn = (size_t)(s + 64 < a.S ? s + 64 : a.S - 1) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { Ln[i] = Lrow[sn + i]; Rn[i] = Rrow[sn + i]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[j] = pg_dot4(L[0], Pl[j], L[1], Pl[4 + j], L[2], Pl[8 + j], L[3], Pl[12 + j]);
            v[j] = pg_dot4(Rv[0], Pr[j], Rv[1], Pr[4 + j], Rv[2], Pr[8 + j], Rv[3], Pr[12 + j]);
        }
        const double x0 = u[0] * v[0], x1 = u[1] * v[1], x2 = u[2] * v[2], x3 = u[3] * v[3];
        const double lik = pg_dot4(pi[0], x0, pi[1], x1, pi[2], x2, pi[3], x3);
        const double inv = alpha * pg_rcp(lik);
        acc[32] = __builtin_fma(x0, inv, acc[32]); acc[33] = __builtin_fma(x1, inv, acc[33]);
        acc[34] = __builtin_fma(x2, inv, acc[34]); acc[35] = __builtin_fma(x3, inv, acc[35]);
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[j] = pi[j] * inv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double tl = xb[j] * v[j], tr = xb[j] * u[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i * 4 + j] = __builtin_fma(L[i], tl, acc[i * 4 + j]);
                acc[16 + i * 4 + j] = __builtin_fma(Rv[i], tr, acc[16 + i * 4 + j]);
            }
        }
    }
    const int q = lane >> 2, part = lane & 3;
    double* out = a.part + node * PG_PART;
#pragma unroll
    for (int b = 0; b < 3; ++b) {                            // the 36 sums over the wave's 64 lanes, fixed order, through its LDS rows
#pragma unroll
        for (int i = 0; i < 12; ++i) red[wv][i][lane] = acc[b * 12 + i];
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
        if (q < 12) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v = v + red[wv][q][part + 4 * i];
        }
        v = v + pg_quad_sum_step<1>(v);
        v = v + pg_quad_sum_step<2>(v);
        if (q < 12 && part == 0) out[b * 12 + q] = v;
        __builtin_amdgcn_wave_barrier();
    }
    // nobody adopted (r, k): G = omega, and every root slot's coefficient with it.  (Last: a store ahead of the uniform loads above
    // turns them into vector loads -- 64 more registers, and the kernel four times slower.)
    if (phase == 0) {
        if (lane == 0) a.G[node] = alpha;
        for (int slot = lane; slot < a.N - r - 1; slot += 64) a.C[node * a.N + slot] = alpha;
    }
}

// Nodes somebody merged again (and, twisted proposal, nodes with look-ahead entries), small nodes (S <= 4096), row form:
// grid (flagged nodes of rank event r: slow_idx[slow0 ...], tiles of 256 sites), thread = site, the node's own matrices in scalar
// registers, whole 32-byte rows per load, fused multiply-adds.  A rank event has a few dozen such nodes, so the launch is a chain
// of latencies, not throughput: every load of a thread is issued as early as possible, nothing goes through memory between the
// adjoint row and the matrix adjoints, and the register count (occupancy 2) does not matter.  The 36 sums of a (node, tile) go
// to slowpart; pg_node_finish adds the tiles in order.
// ALL = false: the flagged nodes of rank event r (a launch per rank event, newest first: a node's flagged parents are complete when
//   its launch starts).
// ALL = true: every flagged node of the sweep in ONE launch, newest rank event first in the grid (workgroups are dispatched in
//   grid order, so what a workgroup waits for was dispatched before it).  A workgroup does everything that does not need its
//   flagged parents -- its rows, matrices, the own term, the chunk sums of the free parents, the staging -- and only then waits
//   for the (parent, same tile) workgroups it gathers from: row_done[parent TS + tile] == epoch, set by the parent's workgroup
//   behind an agent-scope release of its adjoint tile.  The wait is bounded (row_timeout).  The chain of ten launches at their
//   latency floors becomes one launch whose dependent part is the flagged parents' gather alone.
template <bool ALL>
__device__ __forceinline__ void pg_nodes_rows_body(const pg_args& a, int r_arg, int slow0, int chunk0) {
    __shared__ double shP[PG_PCHUNK][32];
    __shared__ int shE[PG_PCHUNK];
    __shared__ int shSib[PG_PCHUNK];
    __shared__ double shA[PG_PCHUNK];
    __shared__ pg_rows_lds sh;
    const int tid = threadIdx.x;
    const int si = ALL ? slow0 - 1 - (int)blockIdx.x : slow0 + (int)blockIdx.x;     // (ALL: slow0 = the number of flagged nodes)
    const size_t node = (size_t)a.slow_idx[si];
    const int r = ALL ? (int)(node / (size_t)a.K) : r_arg;
    const bool has_x = (a.slow_flag[node] & 2) != 0;        // pg_twist_xsum left the look-ahead merges' share in the adjoint row
    const size_t row = (size_t)a.S * 4;
    const int s = blockIdx.y * 256 + tid;
    const bool live = s < a.S;
    const size_t so = (size_t)(live ? s : a.S - 1) * 4;
    const double* Lrow = pg_row(a, a.child[node * 2]);
    const double* Rrow = pg_row(a, a.child[node * 2 + 1]);
    const double* xrow = a.pool + node * row;
    double* orow = a.adj + node * row;
    double x[4], L[4], Rv[4], xb[4], x0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = xrow[so + i]; L[i] = Lrow[so + i]; Rv[i] = Rrow[so + i]; x0[i] = has_x ? orow[so + i] : 0.0; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    const int pbeg = a.par_off[node], pend = a.par_off[node + 1];
    const int hv = a.heavy_first[node];
    const int np = pend - pbeg;
    const int nch = hv >= 0 ? (np + PG_HCHUNK - 1) / PG_HCHUNK : 0;
    // the entries gathered here, PG_PCHUNK at a time through LDS: a light node's whole list; of a heavy node (its free parents'
    // entries are in the chunk sums) the flagged parents' entries, which end the list -- counted by a ballot over its last entries
    int tail_beg = pbeg, tail_n = hv < 0 ? np : 0;
    if (hv >= 0 && a.chunks_free_only) {
        int cnt = 0;
        for (int base = pend;; base -= 64) {
            const int i = base - 1 - (tid & 63);
            const bool fr = i < pbeg || (a.par_idx[i] & PG_FREE_PARENT) != 0;
            const unsigned long long m = __ballot(fr);
            const int run = m ? __ffsll((long long)m) - 1 : 64;
            cnt += run;
            if (run < 64) break;
        }
        tail_n = cnt;
        tail_beg = pend - cnt;
    }
    if (tail_n > 0) pg_stage_parents(a, tail_beg, tail_n < PG_PCHUNK ? tail_n : PG_PCHUNK, shP, shE, shSib, shA);
    const double* Pu = a.Pmat + node * 32;
    double Pl[16], Pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Pl[i] = Pu[i]; Pr[i] = Pu[16 + i]; }
    __syncthreads();
    double acc[PG_PART];
#pragma unroll
    for (int i = 0; i < PG_PART; ++i) acc[i] = 0.0;
    const double lik = pg_dot4(pi[0], x[0], pi[1], x[1], pi[2], x[2], pi[3], x[3]);
    if constexpr (ALL) {                                     // the launch runs beside the coefficient chain: rank event r's must be complete
        if (a.coeff_done && ((a.coeff_mask >> r) & 1ull)) {  // (uniform)
            if (tid == 0) {
                unsigned int spins = 0;
                while (__hip_atomic_load(a.coeff_done + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.row_epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 18)) {
                        __hip_atomic_store(a.row_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
    const double alpha = ALL ? pg_ld_agent(a.C + node * a.N + (a.N - r - 2)) : a.C[node * a.N + (a.N - r - 2)];
    const double inv = alpha * pg_rcp(lik);
#pragma unroll
    for (int j = 0; j < 4; ++j) xb[j] = pi[j] * inv + x0[j];
    if (hv >= 0) {                                           // chunk sums, sixteen rows in flight (added in order)
        const double* cp = a.cpart + (size_t)(chunk0 + hv) * row + so;   // (chunk0: the rank event's first chunk when cpart holds all events')
        for (int c0 = 0; c0 < nch; c0 += 16) {
            double q[16][4];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cc = c0 + u < nch ? c0 + u : nch - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) q[u][j] = cp[(size_t)cc * row + j];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c0 + u < nch) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xb[j] = xb[j] + q[u][j];
                }
        }
    }
    for (int st = 0; st < tail_n; st += PG_PCHUNK) {
        const int nc = tail_n - st < PG_PCHUNK ? tail_n - st : PG_PCHUNK;
        if (st > 0) {                                        // (more than PG_PCHUNK flagged parents: rare)
            __syncthreads();
            pg_stage_parents(a, tail_beg + st, nc, shP, shE, shSib, shA);
            __syncthreads();
        }
        if constexpr (ALL) {                                 // the flagged parents of this batch: their tiles must be complete
            if (tid < nc) {
                const int enc = shE[tid];
                if (!(enc & PG_FREE_PARENT)) {
                    const unsigned int* f = a.row_done + (size_t)(enc >> 1) * a.TS + blockIdx.y;
                    unsigned int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.row_epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 18)) {                 // (~0.3 s; a wait lasts microseconds)
                            __hip_atomic_store(a.row_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        for (int e0 = 0; e0 < nc; e0 += 4) {                 // four at a time, loads first
            double xp[4][4], sb[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e0 + u < nc ? e0 + u : nc - 1;
                const int enc = __builtin_amdgcn_readfirstlane(shE[ee]);
                const double* sbp = pg_row(a, shSib[ee]) + so;
#pragma unroll
                for (int j = 0; j < 4; ++j) { xp[u][j] = 0.0; sb[u][j] = sbp[j]; }
                if (!(enc & PG_FREE_PARENT)) {
                    const double* xpp = a.adj + (size_t)(enc >> 1) * row + so;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xp[u][j] = ALL ? pg_ld_agent(xpp + j) : xpp[j];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u < nc) {
                    const int e = e0 + u;
                    const int enc = __builtin_amdgcn_readfirstlane(shE[e]);
                    const int side = enc & 1;
                    const double* Psib = shP[e] + (1 - side) * 16;
                    const double* Pme = shP[e] + side * 16;
                    double t[4], w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        w[j] = pg_dot4(sb[u][0], Psib[j], sb[u][1], Psib[4 + j], sb[u][2], Psib[8 + j], sb[u][3], Psib[12 + j]);
                    if (enc & PG_FREE_PARENT) {              // the parent's own term, recomputed: alpha_p pi / (pi . ((me P_me) o (sib P_sib)))
                        double lik = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            lik = __builtin_fma(pi[j] * w[j], pg_dot4(x[0], Pme[j], x[1], Pme[4 + j], x[2], Pme[8 + j], x[3], Pme[12 + j]), lik);
                        const double ai = shA[e] * pg_rcp(lik);
#pragma unroll
                        for (int j = 0; j < 4; ++j) xp[u][j] = pi[j] * ai;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = xp[u][j] * w[j];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xb[j] = xb[j] + pg_dot4(t[0], Pme[j * 4], t[1], Pme[j * 4 + 1], t[2], Pme[j * 4 + 2], t[3], Pme[j * 4 + 3]);
                }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ALL) pg_st_agent(orow + so + j, xb[j]); else orow[so + j] = xb[j];
            acc[32 + j] = x[j] * inv;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double u = pg_dot4(L[0], Pl[j], L[1], Pl[4 + j], L[2], Pl[8 + j], L[3], Pl[12 + j]);
            const double v = pg_dot4(Rv[0], Pr[j], Rv[1], Pr[4 + j], Rv[2], Pr[8 + j], Rv[3], Pr[12 + j]);
            const double tl = xb[j] * v, tr = xb[j] * u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i * 4 + j] = L[i] * tl;
                acc[16 + i * 4 + j] = Rv[i] * tr;
            }
        }
    }
    if constexpr (ALL) {                                     // this tile of the adjoint row is complete: release it, then say so
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave's agent-scope stores are acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.row_done + node * a.TS + blockIdx.y, a.row_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pg_rows_reduce(acc, sh, a.slowpart + ((size_t)si * gridDim.y + blockIdx.y) * PG_PART);
}
__global__ __launch_bounds__(256, 2) void pg_nodes_rows(pg_args a, int r, int slow0, int chunk0) { pg_nodes_rows_body<false>(a, r, slow0, chunk0); }
__global__ __launch_bounds__(256, 2) void pg_nodes_rows_all(pg_args a, int n_slow) { pg_nodes_rows_body<true>(a, 0, n_slow, 0); }

// ---- g6: per node: tiles -> Pl_bar, Pr_bar -> branch adjoints and the Q adjoint ----------------------------
__global__ __launch_bounds__(256) void pg_node_finish(pg_args a) {
    // four lanes per (node, side) -- lane i has row i of the Frechet series (pg_expm4_frechet_row: a quarter of one lane's ~5 k
    // dependent instructions; the launch lasts as long as its longest lane: 21 us with a lane per matrix, 36 us with a lane per
    // node) -- 32 nodes per workgroup of 256 threads, grid ceil(R K / 32)
    __shared__ double shw[4][20];
    const int tid = threadIdx.x, row = tid & 3, side = (tid >> 2) & 1;
    const size_t node0 = (size_t)blockIdx.x * 32 + (tid >> 3);
    const bool valid = node0 < (size_t)a.R * a.K;
    const size_t node = valid ? node0 : (size_t)a.R * a.K - 1;
    double pb[16], pib[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; ++q) pb[q] = 0.0;
    const int sf = a.slowpart ? a.slow_flag[node] : 0;
    const int nt = sf ? a.TS : a.T;
    const double* p0 = sf ? a.slowpart + (size_t)(sf >> 3) * a.TS * PG_PART : a.part + node * a.T * PG_PART;
    for (int t = 0; t < nt; ++t) {
        const double* p = p0 + (size_t)t * PG_PART;
#pragma unroll
        for (int q = 0; q < 16; ++q) pb[q] = pb[q] + p[side * 16 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) pib[q] = pib[q] + p[32 + q];
    }
    double Q[16], Pm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Q[i] = a.Q[i];
    const double* P = a.Pmat + node * 32 + side * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) Pm[i] = P[i];
    {                                                        // dP/db = Q P: this lane's row of it, its share of <Pbar, Q P>, then the quad
        const double qrow[4] = {Q[row * 4], Q[row * 4 + 1], Q[row * 4 + 2], Q[row * 4 + 3]};
        double qp[4];
        pg_rowmat(qrow, Pm, qp);
        double bb = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) bb = bb + (row == 0 ? pb[j] : row == 1 ? pb[4 + j] : row == 2 ? pb[8 + j] : pb[12 + j]) * qp[j];
        bb = ((pg_quad<0>(bb) + pg_quad<1>(bb)) + pg_quad<2>(bb)) + pg_quad<3>(bb);
        if (valid && row == 0) a.nodeg[node * PG_NODEG + side] = bb;
    }
    double dQ[4] = {0.0, 0.0, 0.0, 0.0};                     // row `row` of this (node, side)'s share of Q_bar
    if (!a.jc) {                                             // <Pbar, L(Qb, E b)> = <b L((Qb)^T, Pbar), E>
        const double b = (side ? a.br : a.bl)[node];
        double At[16], Lr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) At[i * 4 + j] = Q[j * 4 + i] * b;
        pg_expm4_frechet_row(At, pb, row, Lr);
#pragma unroll
        for (int j = 0; j < 4; ++j) dQ[j] = b * Lr[j];
    }
    // Q_bar and pi_bar are only ever summed over all nodes: entry by entry over the wave's 8 nodes (a fixed butterfly; a lane
    // contributes to the four entries of its row), then the four waves in order, to fin_part[workgroup][20], which pg_reduce adds up
    // -- not 20 strided columns of R K nodes (28 us)
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        double v = 0.0;
        if (i < 16) v = (i >> 2) == row ? dQ[i & 3] : 0.0;
        else v = (side == 0 && row == 0) ? pib[i - 16] : 0.0;
        v = pk_wave_tree_sum(valid ? v : 0.0);               // (DPP and readlane; twenty butterflies through ds_bpermute: 20.3 against 17.9 us)
        if ((tid & 63) == 0) shw[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 20) a.fin_part[(size_t)blockIdx.x * 20 + tid] = ((shw[0][tid] + shw[1][tid]) + shw[2][tid]) + shw[3][tid];
}

// ---- g7: explicit occurrences of b and lambda in ll_r and in the proposal term; pathwise db/dlambda ---------
__global__ __launch_bounds__(256) void pg_scalars(pg_args a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.R * a.K) return;
    const int r = t / a.K, k = t - r * a.K;
    const double om = a.om[t], g = a.G[t];
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const double* lam = side ? a.lam_r : a.lam_l;
        const double* b = side ? a.br : a.bl;
        double suffix = 0.0, prefix = 0.0;
        for (int j = r; j < a.R; ++j) suffix = suffix + a.G[(size_t)j * a.K + k] * lam[j];
        for (int j = 0; j <= r; ++j) prefix = prefix + b[(size_t)j * a.K + k];
        const double lr = lam[r], br_ = b[t];
        const double bbar = (a.nodeg[(size_t)t * PG_NODEG + side] - suffix) + om * lr;
        double term = (g * ((double)(r + 1) / lr - prefix) - om * (1.0 / lr - br_)) + bbar * (-br_ / lr);
        if (a.twist) term = term + a.tw.twnode[(size_t)t * PG_NODEG + side];
        a.terms[(size_t)t * 2 + side] = term;
    }
}

// ---- g8: final sums (fixed order): block o < 2R: d_lam; then d_pi[4], d_Q[16] -------------------------------
__global__ __launch_bounds__(256) void pg_reduce(pg_args a) {
    __shared__ double sh[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    if (o < 2 * a.R) {
        const int side = o / a.R, r = o - side * a.R;
        for (int k = tid; k < a.K; k += 256) acc = acc + a.terms[((size_t)r * a.K + k) * 2 + side];
    } else {
        const int q = o - 2 * a.R;                           // 0..3 pi, 4..19 Q
        const int col = q < 4 ? 18 + q : 2 + (q - 4);
        const int fcol = q < 4 ? 16 + q : q - 4;             // pg_node_finish's per-workgroup sums: Q_bar[16], pi_bar[4]
        const size_t nf = ((size_t)a.R * a.K + 31) / 32;
        for (size_t i = tid; i < nf; i += 256) acc = acc + a.fin_part[i * 20 + fcol];
        if (a.twist) {
            const size_t n = (size_t)a.R * a.K;
            // eight loads in flight (clamped index), added in order: one load per trip of a plain loop costs a latency each
            const double* src = a.tw.twnode;
            for (size_t i0 = tid; i0 < n; i0 += 8 * 256) {
                double q8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t i = i0 + (size_t)u * 256 < n ? i0 + (size_t)u * 256 : n - 1;
                    q8[u] = src[i * PG_NODEG + col];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + (size_t)u * 256 < n) acc = acc + q8[u];
            }
        }
        if (q < 4)
            for (int k = tid; k < a.K; k += 256) acc = acc + a.leafterm[k * 4 + q];
    }
    const double t = pg_block_sum(acc, sh);
    if (tid == 0) a.out[o] = t;
}


// ================================================================================================================
// The twisted proposal (vncsmc.py:295-416): the potentials of EVERY (pair, sub-sample) are differentiated.
//   pg_twist_tau      tau = omega (softmax_j(pot) - [j chosen]) and the root-slot coefficients ctw
//   pg_twist_pbar     per row: Pl_bar, Pr_bar, pi_bar of the look-ahead merge (one wave per row)
//   pg_twist_finish   per particle: branch adjoints -> rate terms, Frechet terms -> Q_bar, of all its rows
//   pg_twist_xchunks  adjoint rows of the adopted roots (internal nodes only), gathered per node in fixed order
//   pg_twist_xsum     chunk sums -> adj[node], which pg_nodes then starts from
// Oracle: oracle/cpu_grad.py sweep_grad_twisted.
// ================================================================================================================
__device__ __forceinline__ int pg_pair_index(int r1, int r2, int n) { return r1 * (2 * n - r1 - 1) / 2 + (r2 - r1 - 1); }

// one wave per (rank event, particle); dynamic LDS: J_0 doubles
__device__ __forceinline__ void pg_twist_tau_body(const pg_args& a, double* w, int t, int r, int k, int n, int M, int J, size_t row0) {
    const int lane = threadIdx.x;
    const double* pot = a.tw.pot + row0;
    double mx = -pm_inf();
    for (int j = lane; j < J; j += 64) {
        const double v = pot[j];
        if (!pm_isnan(v) && v > mx) mx = v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    const bool all_bad = !(mx > -pm_inf()) || mx == pm_inf();   // the weights pk_twist_choose drew from
    double ssum = 0.0;
    for (int j = lane; j < J; j += 64) {
        const double v = pot[j];
        const double wj = all_bad ? 1.0 : (pm_isnan(v) ? 0.0 : pm_exp(v - mx));
        w[j] = wj;
        ssum = ssum + wj;
    }
    ssum = pg_wave_sum(ssum);
    const double om = a.om[t];
    const int jsel = (int)a.tw.chosen[t];
    const double inv = 1.0 / ssum;
    for (int j = lane; j < J; j += 64) {
        const double tj = om * (w[j] * inv - (j == jsel ? 1.0 : 0.0));
        w[j] = tj;
        a.tw.tau[row0 + j] = tj;
    }
    __syncthreads();
    for (int x = lane; x < n; x += 64) {
        double acc = 0.0;
        for (int y = 0; y < n; ++y) {
            if (y == x) continue;
            const int tp = x < y ? pg_pair_index(x, y, n) : pg_pair_index(y, x, n);
            for (int m = 0; m < M; ++m) acc = acc + w[tp * M + m];
        }
        a.tw.ctw[(size_t)t * a.N + x] = -acc;
    }
}
__global__ __launch_bounds__(64) void pg_twist_tau(pg_args a) {
    extern __shared__ __attribute__((aligned(16))) char pg_smem[];
    const int t = blockIdx.x;
    const int r = t / a.K, k = t - r * a.K;
    const int n = a.N - r, M = a.tw.M, J = (n * (n - 1) / 2) * M;
    const size_t row0 = (size_t)a.tw.joff[r] + (size_t)k * J;
    // the weights / tau of the particle's J rows: LDS, or (J beyond 8192) the tau rows themselves as the working array; two
    // instantiations, so that the LDS case keeps LDS instructions
    if (J <= 8192) pg_twist_tau_body(a, reinterpret_cast<double*>(pg_smem), t, r, k, n, M, J, row0);
    else pg_twist_tau_body(a, a.tw.tau + row0, t, r, k, n, M, J, row0);
}

struct pg_rowid { int r, k, j, n, J; };
__device__ __forceinline__ pg_rowid pg_twist_row_of(const pg_args& a, int64_t row) {
    pg_rowid o;
    o.r = 0;
    while (row >= a.tw.joff[o.r + 1]) ++o.r;
    o.n = a.N - o.r;
    o.J = (o.n * (o.n - 1) / 2) * a.tw.M;
    const int64_t rel = row - a.tw.joff[o.r];
    o.k = (int)(rel / o.J);
    o.j = (int)(rel - (int64_t)o.k * o.J);
    return o;
}
__device__ __forceinline__ void pg_pair_of(int t, int n, int& il, int& ir) {
    il = 0;
    int rem = t;
    while (rem >= n - 1 - il) { rem -= n - 1 - il; ++il; }
    ir = il + 1 + rem;
}

// One site of a look-ahead merge: acc += w * { X1^T (g o v),  X2^T (g o u),  y / lik }  with u = X1 Pl, v = X2 Pr, y = u o v,
// lik = pi . y, g = pi / lik  (the factor tau is applied by pg_twist_finish)
__device__ __forceinline__ void pg_pbar_site(const double (&x1)[4], const double (&x2)[4], const double (&Pl)[16],
                                             const double (&Pr)[16], const double (&pi)[4], double w, double (&acc)[PG_PART]) {
    double u[4], v[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[j] = pg_dot4(x1[0], Pl[j], x1[1], Pl[4 + j], x1[2], Pl[8 + j], x1[3], Pl[12 + j]);
        v[j] = pg_dot4(x2[0], Pr[j], x2[1], Pr[4 + j], x2[2], Pr[8 + j], x2[3], Pr[12 + j]);
        y[j] = u[j] * v[j];
    }
    const double lik = pg_dot4(pi[0], y[0], pi[1], y[1], pi[2], y[2], pi[3], y[3]);
    const double inv = w * pg_rcp(lik);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double g = pi[j] * inv;
        const double gv = g * v[j], gu = g * u[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i * 4 + j] = __builtin_fma(x1[i], gv, acc[i * 4 + j]);
            acc[16 + i * 4 + j] = __builtin_fma(x2[i], gu, acc[16 + i * 4 + j]);
        }
        acc[32 + j] = __builtin_fma(y[j], inv, acc[32 + j]);
    }
}

// one wave per row (4 rows per workgroup).  Rows of two coded leaves are left to pg_twist_pbar_ll.
// The 36 sums over the wave's 64 lanes go through LDS, 12 at a time: lane (q, part) adds 16 of the 64 values of sum q.
__global__ __launch_bounds__(256, 4) void pg_twist_pbar(pg_args a) {
    __shared__ double red[4][12][PG_RED_STRIDE];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= a.tw.joff[a.R]) return;
    const pg_rowid id = pg_twist_row_of(a, row);
    int il, ir;
    pg_pair_of(id.j / a.tw.M, id.n, il, ir);
    const int32_t* ro = a.tw.roots_ad + ((size_t)id.r * a.K + id.k) * a.N;
    const int n1 = ro[il], n2 = ro[ir];
    if (a.tw.pair_hist && n1 < a.N && n2 < a.N) return;
    const double* X1 = pg_row(a, n1);
    const double* X2 = pg_row(a, n2);
    const double* P = a.tw.tw_P + (size_t)row * 32;
    double Pl[16], Pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Pl[i] = P[i]; Pr[i] = P[16 + i]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double acc[PG_PART];
#pragma unroll
    for (int i = 0; i < PG_PART; ++i) acc[i] = 0.0;
    for (int s = lane; s < a.S; s += 64) {
        double x1[4], x2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x1[i] = X1[(size_t)s * 4 + i]; x2[i] = X2[(size_t)s * 4 + i]; }
        pg_pbar_site(x1, x2, Pl, Pr, pi, 1.0, acc);
    }
    double* out = a.tw.twpart + (size_t)row;
    const size_t nrows = (size_t)a.tw.joff[a.R];
    const int q = lane >> 2, part = lane & 3;
#pragma unroll
    for (int b = 0; b < 3; ++b) {                            // a wave's own LDS rows: no workgroup barrier needed
#pragma unroll
        for (int i = 0; i < 12; ++i) red[wv][i][lane] = acc[b * 12 + i];
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
        if (q < 12) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v = v + red[wv][q][part + 4 * i];
        }
        v = v + pg_quad_sum_step<1>(v);
        v = v + pg_quad_sum_step<2>(v);
        if (q < 12 && part == 0) out[(size_t)(b * 12 + q) * nrows] = v;
        __builtin_amdgcn_wave_barrier();
    }
}

// rows of two coded leaves: the site terms take one of 25 values, pair_hist holds how many sites take each.  One thread per
// row of rank event r: grid (ceil(K J_r / 64)), 64 threads
__global__ __launch_bounds__(64) void pg_twist_pbar_ll(pg_args a, int r) {
    const int n = a.N - r, M = a.tw.M, J = (n * (n - 1) / 2) * M;
    const int64_t rel = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (rel >= (int64_t)a.K * J) return;
    const int k = (int)(rel / J), j = (int)(rel - (int64_t)k * J);
    int il, ir;
    pg_pair_of(j / M, n, il, ir);
    const int32_t* ro = a.tw.roots_ad + ((size_t)r * a.K + k) * a.N;
    const int n1 = ro[il], n2 = ro[ir];
    if (n1 >= a.N || n2 >= a.N) return;
    const size_t row = (size_t)a.tw.joff[r] + (size_t)rel;
    const double* P = a.tw.tw_P + row * 32;
    double Pl[16], Pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Pl[i] = P[i]; Pr[i] = P[16 + i]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double acc[PG_PART];
#pragma unroll
    for (int i = 0; i < PG_PART; ++i) acc[i] = 0.0;
    const uint32_t* hist = a.tw.pair_hist + ((size_t)n1 * a.N + n2) * 32;
#pragma unroll 1
    for (int c = 0; c < 25; ++c) {
        const uint32_t cnt = hist[c];
        if (cnt == 0) continue;
        const int cl = c / 5, cr = c - cl * 5;
        double x1[4], x2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x1[i] = (cl == 4 || cl == i) ? 1.0 : 0.0;
            x2[i] = (cr == 4 || cr == i) ? 1.0 : 0.0;
        }
        pg_pbar_site(x1, x2, Pl, Pr, pi, (double)cnt, acc);
    }
    const size_t nrows = (size_t)a.tw.joff[a.R];
#pragma unroll
    for (int i = 0; i < PG_PART; ++i) a.tw.twpart[(size_t)i * nrows + row] = acc[i];
}

// Rows of rank event r -> branch adjoints -> rate terms, Frechet terms -> Q adjoint, summed per particle.  A workgroup takes
// KB = max(1, 256 / J_r) particles (their KB J_r rows are contiguous), one thread per row; the 22 results of a row go to LDS and
// thread (particle, result) adds the particle's J_r rows in order.  grid (ceil(K / KB)), 256 threads, LDS 22 x 256 doubles.
__global__ __launch_bounds__(256) void pg_twist_finish(pg_args a, int r) {
    __shared__ double sh[PG_NODEG][256];
    const int n = a.N - r, J = (n * (n - 1) / 2) * a.tw.M;
    const int KB = J >= 256 ? 1 : 256 / J;
    const int k0 = blockIdx.x * KB;
    const int kb = a.K - k0 < KB ? a.K - k0 : KB;            // particles of this workgroup
    const size_t row0 = (size_t)a.tw.joff[r] + (size_t)k0 * J;
    // J > 256 (many sub-samples): gridDim.y slices of 256 rows per particle, summed afterwards by pg_twist_finish_sum -- with few
    // particles (the K = 32..64 of the reference's experiments) one workgroup per particle walking 660 rows left the GPU empty
    const int nsl = (int)gridDim.y, sl = (int)blockIdx.y;
    const double ll = a.lam_l[r], lr = a.lam_r[r];
    double Q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Q[i] = a.Q[i];
    double tot[PG_NODEG];                                     // J > 256: a thread keeps the sum of its rows j = tid, tid + 256, ...
#pragma unroll
    for (int i = 0; i < PG_NODEG; ++i) tot[i] = 0.0;
    const int nrows = nsl > 1 ? (J < (sl + 1) * 256 ? J : (sl + 1) * 256) : kb * J;
    for (int t0 = sl * 256; t0 < (nsl > 1 ? (sl + 1) * 256 : (J >= 256 ? J : 256)); t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        double res[PG_NODEG];
#pragma unroll
        for (int i = 0; i < PG_NODEG; ++i) res[i] = 0.0;
        const size_t row = row0 + t;
        const double tau = t < nrows ? a.tw.tau[row] : 0.0;
        if (tau != 0.0) {
            const double* pp = a.tw.twpart + row;
            const size_t nrows_all = (size_t)a.tw.joff[a.R];
#pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                double Pm[16], QP[16], pb[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { Pm[i] = a.tw.tw_P[row * 32 + side * 16 + i]; pb[i] = tau * pp[(size_t)(side * 16 + i) * nrows_all]; }
                pm_mm4(Q, Pm, QP);
                double bb = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) bb = bb + pb[i] * QP[i];
                const double b = a.tw.tw_b[row * 2 + side];
                res[side] = bb * (-b / (side ? lr : ll));
                if (!a.jc) {
                    double At[16], Lf[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) At[i * 4 + jj] = Q[jj * 4 + i] * b;
                    pg_expm4_frechet(At, pb, Lf);
#pragma unroll
                    for (int i = 0; i < 16; ++i) res[2 + i] = res[2 + i] + b * Lf[i];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) res[18 + q] = tau * pp[(size_t)(32 + q) * nrows_all];
        }
#pragma unroll
        for (int i = 0; i < PG_NODEG; ++i) tot[i] = tot[i] + res[i];
    }
#pragma unroll
    for (int i = 0; i < PG_NODEG; ++i) sh[i][threadIdx.x] = tot[i];
    __syncthreads();
    const int span = J >= 256 ? 256 : J;                      // LDS entries per particle
    for (int o = threadIdx.x; o < kb * PG_NODEG; o += 256) {
        const int kl = o / PG_NODEG, v = o - kl * PG_NODEG;
        double acc = 0.0;
        for (int j = 0; j < span; ++j) acc = acc + sh[v][kl * span + j];
        if (nsl > 1) a.tw.twslice[((size_t)(k0 + kl) * nsl + sl) * PG_NODEG + v] = acc;
        else a.tw.twnode[((size_t)r * a.K + k0 + kl) * PG_NODEG + v] = acc;
    }
}
__global__ __launch_bounds__(256) void pg_twist_finish_sum(pg_args a, int r, int nsl) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.K * PG_NODEG) return;
    const int k = o / PG_NODEG, v = o - k * PG_NODEG;
    double acc = 0.0;
    for (int s = 0; s < nsl; ++s) acc = acc + a.tw.twslice[((size_t)k * nsl + s) * PG_NODEG + v];
    a.tw.twnode[((size_t)r * a.K + k) * PG_NODEG + v] = acc;
}

// grid (chunks of rank event r, groups of 256 sites): a thread owns one site of the chunk's node x and walks the chunk's
// (adopter, slot) entries; for each, every partner slot and sub-sample:  xb += (tau g o v) Pme^T
__global__ __launch_bounds__(256) void pg_twist_xchunks(pg_args a, int r, int chunk0) {
    const int ci = blockIdx.x, c = chunk0 + ci;
    const int x = a.tw.xchunk_node[c], beg = a.tw.xchunk_beg[c], cnt = a.tw.xchunk_cnt[c];
    const int p0 = a.tw.xchunk_part[c] & 0xffff, p1 = a.tw.xchunk_part[c] >> 16;
    const int s = blockIdx.y * 256 + threadIdx.x;
    const bool live = s < a.S;
    const size_t soff = (size_t)(live ? s : a.S - 1) * 4;
    const int n = a.N - r, M = a.tw.M, J = (n * (n - 1) / 2) * M;
    const double* xr = a.pool + (size_t)(x - a.N) * a.S * 4 + soff;
    const double x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3];
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double xb[4] = {0.0, 0.0, 0.0, 0.0};
    for (int e = 0; e < cnt; ++e) {
        const int enc = a.tw.xent[beg + e];
        const int kp = enc / a.N, i = enc - kp * a.N;
        const int32_t* ro = a.tw.roots_ad + ((size_t)r * a.K + kp) * a.N;
        const size_t row0 = (size_t)a.tw.joff[r] + (size_t)kp * J;
        for (int i2 = p0; i2 < p1; ++i2) {
            if (i2 == i) continue;
            const int side = i > i2 ? 1 : 0;                  // x is the right child of the look-ahead merge
            const int tp = side ? pg_pair_index(i2, i, n) : pg_pair_index(i, i2, n);
            const double* sr = pg_row(a, ro[i2]) + soff;
            const double s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3];
            for (int m = 0; m < M; ++m) {
                const size_t row = row0 + (size_t)tp * M + m;
                const double tau = a.tw.tau[row];
                const double* Pme = a.tw.tw_P + row * 32 + side * 16;
                const double* Psb = a.tw.tw_P + row * 32 + (1 - side) * 16;
                // with z = Pme (pi o v):  lik = x . z  and the contribution (tau (pi o v) / lik) Pme^T = tau z / lik
                double z[4];
                {
                    double pv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pv[j] = pi[j] * pg_dot4(s0, Psb[j], s1, Psb[4 + j], s2, Psb[8 + j], s3, Psb[12 + j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        z[q] = pg_dot4(pv[0], Pme[q * 4], pv[1], Pme[q * 4 + 1], pv[2], Pme[q * 4 + 2], pv[3], Pme[q * 4 + 3]);
                }
                const double lik = pg_dot4(x0, z[0], x1, z[1], x2, z[2], x3, z[3]);
                const double f = tau * pg_rcp(lik);
#pragma unroll
                for (int q = 0; q < 4; ++q) xb[q] = __builtin_fma(f, z[q], xb[q]);
            }
        }
    }
    if (live) {
        double* out = a.tw.tpart + ((size_t)ci * a.S + s) * 4;
        out[0] = xb[0]; out[1] = xb[1]; out[2] = xb[2]; out[3] = xb[3];
    }
}

// grid (nodes of rank event r that have entries, groups of 256 elements of a node row)
__global__ __launch_bounds__(256) void pg_twist_xsum(pg_args a, int node0, int chunk0) {
    const int ni = node0 + blockIdx.x;
    const int x = a.tw.xnode_id[ni], c0 = a.tw.xnode_chunk0[ni] - chunk0, ncf = a.tw.xnode_nchunks[ni];
    const int nc = ncf & 0x3fffffff;                         // bit 30: the node's first visit (newest rank event): nothing to add to
    const size_t e = (size_t)blockIdx.y * 256 + threadIdx.x, row = (size_t)a.S * 4;
    if (e >= row) return;
    double* dst = a.adj + (size_t)(x - a.N) * row + e;
    double v = (ncf >> 30) ? 0.0 : *dst;
    const double* src = a.tw.tpart + (size_t)c0 * row + e;
    for (int cb = 0; cb < nc; cb += 8) {                     // eight chunk rows in flight, added in order
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = src[(size_t)(cb + u < nc ? cb + u : nc - 1) * row];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (cb + u < nc) v = v + q[u];
    }
    *dst = v;
}
