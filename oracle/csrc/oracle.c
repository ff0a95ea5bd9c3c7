/* oracle.c -- CPU oracle of the Felsenstein-pruning likelihood + CSMC particle loop, plain C.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
 * never by anything under phylo_amd/.
 *
 * The sweep (ora_sweep) is written in the REFERENCE'S OWN DATAFLOW, function by function from
 * vcsmc.py read as text: a K-replicated core tensor [K, n, S, 4] (vcsmc.py:479), a full gather of it
 * on every resampling (vcsmc.py:286), the three gather_across_core copies of every rank event
 * (vcsmc.py:361-365), and compute_forest_posterior over ALL roots at every rank event
 * (vcsmc.py:231-245, 376).  That is what makes it usable as the timed "reference CPU path" next to
 * the GPU (bench.py cpu_baseline, kind "port"); the HIP path reaches the same numbers without the
 * copies (node pool + integer root tables).
 *
 * Arithmetic follows ora_math.h, so results are bit-identical to the HIP path by construction.
 * Pinned against oracle/cpu_ref.py (NumPy, itself pinned to csmc.py golden vectors) in
 * tests/test_oracle_c.py.  Build: oracle/build.sh (gcc -O2 -fopenmp -ffp-contract=off -mfma).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ora_math.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_QUIRK_Q1_RAW_Q 1u

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* contract v5: the site tile of the canonical sum over sites.  0 restores the policy ora_site_tile(S). */
void ora_set_site_tile(int T) { ora_tile_override = T > 0 ? T : 0; }
int ora_get_site_tile(long S) { return ora_site_tile(S); }

void ora_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* op 0 exp, 1 log, 2 x/y, 3 fma(x,y,x) */
void ora_math_probe(int op, const double* x, const double* y, int n, double* out) {
    for (int i = 0; i < n; ++i) {
        switch (op) {
            case 0: out[i] = ora_exp(x[i]); break;
            case 1: out[i] = ora_log(x[i]); break;
            case 2: out[i] = x[i] / y[i]; break;
            default: out[i] = ora_fma(x[i], y[i], x[i]); break;
        }
    }
}

void ora_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, uint32_t* out4) {
    ora_philox(c0, c1, c2, c3, seed, out4);
}

/* tf.linalg.expm(tensordot(t, Q, 0)), vcsmc.py:181-184 */
void ora_expm_batched(const double* Q, const double* t, int n, int jc, double* P) {
    for (int i = 0; i < n; ++i) {
        if (jc) ora_jc69(t[i], P + (size_t)i * 16); else ora_expm4(Q, t[i], P + (size_t)i * 16);
    }
}

/* one site of vcsmc.py:185-187 (L . P, quirk Q2) */
static inline void merge_site(const double* L, const double* R, const double* Pl, const double* Pr, double* out) {
    for (int j = 0; j < 4; ++j) {
        double lp = L[0] * Pl[j];
        lp = ora_fma(L[1], Pl[4 + j], lp);
        lp = ora_fma(L[2], Pl[8 + j], lp);
        lp = ora_fma(L[3], Pl[12 + j], lp);
        double rp = R[0] * Pr[j];
        rp = ora_fma(R[1], Pr[4 + j], rp);
        rp = ora_fma(R[2], Pr[8 + j], rp);
        rp = ora_fma(R[3], Pr[12 + j], rp);
        out[j] = lp * rp;
    }
}

static inline double site_lik(const double* pi, const double* x) {
    double a = pi[0] * x[0];
    a = ora_fma(pi[1], x[1], a);
    a = ora_fma(pi[2], x[2], a);
    a = ora_fma(pi[3], x[3], a);
    return a;
}

/* sum_s log(pi . x[s]) in the canonical order */
static double row_loglik(const double* pi, const double* x, int S) {
    ora_canon_lp c;
    ora_canon_lp_init(&c, S);
    for (int s = 0; s < S; ++s) ora_canon_lp_mul(&c, s, site_lik(pi, x + (size_t)s * 4));
    return ora_canon_lp_total(&c);
}

/* VCSMC.broadcast_conditional_likelihood_K, vcsmc.py:180-188 */
void ora_cond_likelihood_K(const double* Q, int jc, const double* l, const double* r, const double* tl,
                           const double* tr, int K, int S, double* out) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        double Pl[16], Pr[16];
        if (jc) { ora_jc69(tl[k], Pl); ora_jc69(tr[k], Pr); }
        else { ora_expm4(Q, tl[k], Pl); ora_expm4(Q, tr[k], Pr); }
        const size_t base = (size_t)k * S * 4;
        for (int s = 0; s < S; ++s) merge_site(l + base + (size_t)s * 4, r + base + (size_t)s * 4, Pl, Pr, out + base + (size_t)s * 4);
    }
}

/* log (2 max(c,2) - 3)!! by the loop of vcsmc.py:30-57 */
static double log_double_factorial_count(int c) {
    int m = 2 * (c > 2 ? c : 2) - 3;
    double res = 0.0;
    for (int v = m; v >= 2; v -= 2) res = res + ora_log((double)v);
    return res;
}

/* VCSMC.compute_forest_posterior, vcsmc.py:231-245 */
void ora_forest_loglik(const double* pi, const double* core, const int32_t* record, int K, int X, int S, double* out) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        double fl = 0.0, fp = 0.0;
        for (int x = 0; x < X; ++x) {
            fl = fl + row_loglik(pi, core + ((size_t)k * X + x) * S * 4, S);
            fp = fp + (-log_double_factorial_count(record[(size_t)k * X + x]));
        }
        out[k] = fl + fp;
    }
}

/* CSMC.compute_log_conditional_likelihood (csmc.py:259-326) on arrays; nodes [n_nodes][S][4] scratch */
int ora_tree_loglik(const double* Q, int jc, int n_nodes, int n_leaves, int S, const int32_t* left, const int32_t* right,
                    const double* bl, const double* br, int root, const double* leaves, const double* prior,
                    double* out_loglik, double* root_data) {
    double* nodes = (double*)malloc((size_t)n_nodes * S * 4 * sizeof(double));
    char* done = (char*)calloc((size_t)n_nodes, 1);
    int* stack = (int*)malloc(((size_t)4 * n_nodes + 16) * sizeof(int));
    if (!nodes || !done || !stack) { free(nodes); free(done); free(stack); return -1; }
    memcpy(nodes, leaves, (size_t)n_leaves * S * 4 * sizeof(double));
    for (int i = 0; i < n_leaves; ++i) done[i] = 2;
    int sp = 0;
    stack[sp++] = root;
    while (sp > 0) {
        int v = stack[sp - 1];
        if (done[v] == 2) { --sp; continue; }
        int lc = left[v], rc = right[v];
        if (done[v] == 0) {
            done[v] = 1;
            stack[sp++] = lc;
            stack[sp++] = rc;
            continue;
        }
        --sp;
        double Pl[16], Pr[16];
        if (jc) { ora_jc69(bl[v], Pl); ora_jc69(br[v], Pr); }
        else { ora_expm4(Q, bl[v], Pl); ora_expm4(Q, br[v], Pr); }
        for (int s = 0; s < S; ++s)
            merge_site(nodes + ((size_t)lc * S + s) * 4, nodes + ((size_t)rc * S + s) * 4, Pl, Pr, nodes + ((size_t)v * S + s) * 4);
        done[v] = 2;
    }
    *out_loglik = row_loglik(prior, nodes + (size_t)root * S * 4, S);
    if (root_data) memcpy(root_data, nodes + (size_t)root * S * 4, (size_t)S * 4 * sizeof(double));
    free(nodes); free(done); free(stack);
    return 0;
}

/* ---- resampling: vcsmc.py:284-285 by the integer-CDF contract ----------------------------------- */
static double weights_prepare(const double* logw, int K, uint64_t* cdf /* may be NULL */) {
    double m = -ORA_INF;
    for (int k = 0; k < K; ++k) if (!ora_isnan(logw[k]) && logw[k] > m) m = logw[k];
    int all_bad = !(m > -ORA_INF) || m == ORA_INF;
    ora_canon c;
    ora_canon_init(&c);
    uint64_t run = 0;
    for (int k = 0; k < K; ++k) {
        double v = logw[k];
        double w = all_bad ? 1.0 : (ora_isnan(v) ? 0.0 : ora_exp(v - m));
        ora_canon_add(&c, k, w);
        if (cdf) {
            uint64_t wi = all_bad ? 1ull : (ora_isnan(v) ? 0ull : (uint64_t)(w * 17592186044416.0));
            run += wi;
            cdf[k] = run;
        }
    }
    double sum = ora_canon_total(&c);
    return ((all_bad ? 0.0 : m) + ora_log(sum)) - ora_log((double)K);   /* logsumexp_k(logw) - log K */
}

static int cdf_search(const uint64_t* cdf, int K, uint64_t thr) {
    int lo = 0, hi = K;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] > thr) hi = mid; else lo = mid + 1;
    }
    return lo < K ? lo : K - 1;
}

void ora_resample(const double* logw, int K, uint64_t seed, uint32_t step, int64_t* idx) {
    uint64_t* cdf = (uint64_t*)malloc((size_t)K * sizeof(uint64_t));
    weights_prepare(logw, K, cdf);
    for (int k = 0; k < K; ++k) {
        uint32_t x[4];
        ora_philox((uint32_t)k, step, 2u /* RESAMPLE */, 0u, seed, x);
        uint64_t R = ((uint64_t)x[1] << 32) | x[0];
        idx[k] = cdf_search(cdf, K, ora_mulhi64(R, cdf[K - 1]));
    }
    free(cdf);
}

/* VCSMC.compute_log_ZSMC, vcsmc.py:270-277 */
double ora_log_zsmc(const double* logw, int R, int K) {
    double z = 0.0;
    for (int r = 0; r < R; ++r) z = z + weights_prepare(logw + (size_t)r * K, K, NULL);
    return z;
}

/* ---- the sweep: vcsmc.py:332-451 in the reference's dataflow ------------------------------------ */
/* nodes_out (may be NULL): [(N-1)][K][S][4], the partial-likelihood vector each particle created at
 * each rank event.  Returns 0, or -1 on allocation failure. */
int ora_sweep(const double* genome /*[N][S][4]*/, const double* Q, const double* pi, const double* lam_l,
              const double* lam_r, int jc, int K, int N, int S, uint64_t seed, uint32_t flags,
              double* log_weights /*[(N-1)][K]*/, double* log_lik, double* lbranch, double* rbranch,
              int32_t* merges /*[(N-1)][K][2]*/, int64_t* ancestors /*[(N-2)][K]*/, double* logZ, double* nodes_out) {
    const int R = N - 1;
    const size_t node = (size_t)S * 4, part = (size_t)N * node;
    double* coreA = (double*)malloc((size_t)K * part * sizeof(double));
    double* coreB = (double*)malloc((size_t)K * part * sizeof(double));
    int32_t* recA = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));
    int32_t* recB = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));
    double* lw = (double*)malloc((size_t)R * K * sizeof(double));
    double* ll = (double*)malloc((size_t)R * K * sizeof(double));
    double* bls = (double*)malloc((size_t)R * K * sizeof(double));
    double* brs = (double*)malloc((size_t)R * K * sizeof(double));
    double* ll_tilde = (double*)malloc((size_t)K * sizeof(double));
    uint64_t* cdf = (uint64_t*)malloc((size_t)K * sizeof(uint64_t));
    int64_t* idx = (int64_t*)malloc((size_t)K * sizeof(int64_t));
    double* lse = (double*)malloc((size_t)R * sizeof(double));
    if (!coreA || !coreB || !recA || !recB || !lw || !ll || !bls || !brs || !ll_tilde || !cdf || !idx || !lse) {
        free(coreA); free(coreB); free(recA); free(recB); free(lw); free(ll); free(bls); free(brs);
        free(ll_tilde); free(cdf); free(idx); free(lse);
        return -1;
    }
    /* data = np.array([genome] * K), vcsmc.py:479; leafnode_num_record = 1, :415; ll_tilde = log(1/K), :422 */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        memcpy(coreA + (size_t)k * part, genome, part * sizeof(double));
        for (int i = 0; i < N; ++i) recA[(size_t)k * N + i] = 1;
    }
    const double ll_tilde0 = ora_log(1.0 / (double)K);
    for (int k = 0; k < K; ++k) ll_tilde[k] = ll_tilde0;

    for (int r = 0; r < R; ++r) {
        const int n = N - r;
        if (r > 0) {                                              /* cond_true_resample, vcsmc.py:318-325 */
            lse[r - 1] = weights_prepare(lw + (size_t)(r - 1) * K, K, cdf);
#pragma omp parallel for schedule(static)
            for (int k = 0; k < K; ++k) {
                uint32_t x[4];
                ora_philox((uint32_t)k, (uint32_t)r, 2u, 0u, seed, x);
                uint64_t Rr = ((uint64_t)x[1] << 32) | x[0];
                int a = cdf_search(cdf, K, ora_mulhi64(Rr, cdf[K - 1]));
                idx[k] = a;
                memcpy(coreB + (size_t)k * part, coreA + (size_t)a * part, (size_t)n * node * sizeof(double));  /* tf.gather(core, indices) */
                memcpy(recB + (size_t)k * N, recA + (size_t)a * N, (size_t)n * sizeof(int32_t));
                ll_tilde[k] = ll[(size_t)(r - 1) * K + a];        /* :322-323 */
                if (ancestors) ancestors[(size_t)(r - 1) * K + k] = a;
            }
            double* tc = coreA; coreA = coreB; coreB = tc;
            int32_t* tr_ = recA; recA = recB; recB = tr_;
        }
        const double laml = lam_l[r], lamr = lam_r[r];
        const double loglaml = ora_log(laml), loglamr = ora_log(lamr);
        const double q = 1.0 / ((double)((n - 1) * n) / 2.0);     /* 1 / ncr(N - r, 2), vcsmc.py:298 */
        const double qterm = (flags & ORA_QUIRK_Q1_RAW_Q) ? q : ora_log(q);
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            /* extend_partial_state, vcsmc.py:291-316: keys -> (left, right), remaining ascending */
            uint32_t key[1024];
            for (int b = 0; b < (n + 3) / 4; ++b) ora_philox((uint32_t)k, (uint32_t)r, 0u, (uint32_t)b, seed, key + b * 4);
            int il = 0;
            for (int i = 1; i < n; ++i) if (key[i] > key[il]) il = i;
            int ir = (il == 0) ? 1 : 0;
            for (int i = 0; i < n; ++i) if (i != il && i != ir && key[i] > key[ir]) ir = i;
            int rem[1024], nrem = 0;
            for (int i = 0; i < n; ++i) if (i != il && i != ir) rem[nrem++] = i;
            for (int i = 1; i < nrem; ++i) {                      /* insertion sort by (key, slot) ascending */
                int v = rem[i], j = i - 1;
                while (j >= 0 && (key[rem[j]] > key[v] || (key[rem[j]] == key[v] && rem[j] > v))) { rem[j + 1] = rem[j]; --j; }
                rem[j + 1] = v;
            }
            /* branch lengths, vcsmc.py:351-358 */
            uint32_t x[4];
            ora_philox((uint32_t)k, (uint32_t)r, 1u, 0u, seed, x);
            const double tl = (-ora_log(ora_unit_oc(x[0], x[1]))) / laml;
            const double tr = (-ora_log(ora_unit_oc(x[2], x[3]))) / lamr;
            bls[(size_t)r * K + k] = tl;
            brs[(size_t)r * K + k] = tr;
            double Pl[16], Pr[16];
            if (jc) { ora_jc69(tl, Pl); ora_jc69(tr, Pr); } else { ora_expm4(Q, tl, Pl); ora_expm4(Q, tr, Pr); }
            /* gather_across_core x3 + concat, vcsmc.py:361-368 */
            const double* src = coreA + (size_t)k * part;
            double* dst = coreB + (size_t)k * part;
            const int32_t* rs = recA + (size_t)k * N;
            int32_t* rd = recB + (size_t)k * N;
            for (int p = 0; p < nrem; ++p) {
                memcpy(dst + (size_t)p * node, src + (size_t)rem[p] * node, node * sizeof(double));
                rd[p] = rs[rem[p]];
            }
            const double* L = src + (size_t)il * node;
            const double* Rr = src + (size_t)ir * node;
            double* nw = dst + (size_t)nrem * node;
            for (int s = 0; s < S; ++s) merge_site(L + (size_t)s * 4, Rr + (size_t)s * 4, Pl, Pr, nw + (size_t)s * 4);
            rd[nrem] = rs[il] + rs[ir];                           /* :370-373 */
            if (nodes_out) memcpy(nodes_out + ((size_t)r * K + k) * node, nw, node * sizeof(double));
            if (merges) { merges[((size_t)r * K + k) * 2] = il; merges[((size_t)r * K + k) * 2 + 1] = ir; }
            /* compute_forest_posterior over ALL n-1 roots, vcsmc.py:376 */
            double fl = 0.0, fp = 0.0;
            int vminus = 0;
            for (int xr = 0; xr < n - 1; ++xr) {
                fl = fl + row_loglik(pi, dst + (size_t)xr * node, S);
                fp = fp + (-log_double_factorial_count(rd[xr]));
                vminus += rd[xr] - (rd[xr] == 1 ? 1 : 0);         /* overcounting_correct, :251 */
            }
            /* branch log-priors over history rows 1..r+1 with THIS rank's rate, :378-384 (quirk Q3) */
            double lp = 0.0, rp = 0.0;
            for (int j = 0; j <= r; ++j) {
                lp = lp + ((-laml) * bls[(size_t)j * K + k] + loglaml);
                rp = rp + ((-lamr) * brs[(size_t)j * K + k] + loglamr);
            }
            const double llr = ((fl + fp) + lp) + rp;
            const double paren = ((loglaml - laml * tl) + loglamr) - lamr * tr;
            const double w = (((llr - ll_tilde[k]) - paren) + ora_log((double)vminus)) - qterm;   /* :390-392 */
            ll[(size_t)r * K + k] = llr;
            lw[(size_t)r * K + k] = w;
        }
        { double* tc = coreA; coreA = coreB; coreB = tc; int32_t* tr_ = recA; recA = recB; recB = tr_; }
    }
    lse[R - 1] = weights_prepare(lw + (size_t)(R - 1) * K, K, NULL);
    double z = 0.0;                                               /* compute_log_ZSMC; row 0 contributes 0 */
    for (int r = 0; r < R; ++r) z = z + lse[r];
    if (logZ) *logZ = z;
    if (log_weights) memcpy(log_weights, lw, (size_t)R * K * sizeof(double));
    if (log_lik) memcpy(log_lik, ll, (size_t)R * K * sizeof(double));
    if (lbranch) memcpy(lbranch, bls, (size_t)R * K * sizeof(double));
    if (rbranch) memcpy(rbranch, brs, (size_t)R * K * sizeof(double));
    free(coreA); free(coreB); free(recA); free(recB); free(lw); free(ll); free(bls); free(brs);
    free(ll_tilde); free(cdf); free(idx); free(lse);
    return 0;
}

/* Leaf codes of the reference's encoding (runner.py:83-96): 0..3 one-hot state, 4 all-ones.  Returns 1 when every
 * leaf row is one of them.  For such alignments the look-ahead potential of a leaf-leaf pair is priced by state
 * pair: sum_s log f(c_l[s], c_r[s]) = sum over the 25 code pairs of count * log f (contract v3);
 * term of code pair c goes to column c of a single 64-column tile. */
static int leaf_codes(const double* genome, size_t rows, uint8_t* codes) {
    for (size_t i = 0; i < rows; ++i) {
        const double* x = genome + i * 4;
        int ones = 0, zeros = 0, last = 0;
        for (int j = 0; j < 4; ++j) {
            if (x[j] == 1.0) { ++ones; last = j; }
            else if (ora_bits(x[j]) == 0) ++zeros;
        }
        if (ones == 1 && zeros == 3) codes[i] = (uint8_t)last;
        else if (ones == 4) codes[i] = 4;
        else return 0;
    }
    return 1;
}
static void code_row(int c, double* v) {
    for (int j = 0; j < 4; ++j) v[j] = (c == 4 || c == j) ? 1.0 : 0.0;
}

/* ---- T: the twisted / nested proposal, vncsmc.py:295-416 + 432-499, same dataflow as ora_sweep -------- */
/* potentials_out (may be NULL): [(N-1)][K][Jmax] raw (un-normalised) look-ahead potentials, Jmax = C(N,2)*M,
 * rows padded with zeros (test surface for the potentials kernel). */
int ora_sweep_twisted(const double* genome, const double* Q, const double* pi, const double* lam_l, const double* lam_r,
                      int jc, int K, int N, int S, int M, uint64_t seed, double* log_weights, double* log_lik,
                      double* lbranch, double* rbranch, int32_t* merges, int64_t* ancestors, double* logZ,
                      double* nodes_out, double* potentials_out) {
    const int R = N - 1;
    const size_t node = (size_t)S * 4, part = (size_t)N * node;
    const int Jmax = (N * (N - 1) / 2) * M;
    double* coreA = (double*)malloc((size_t)K * part * sizeof(double));
    double* coreB = (double*)malloc((size_t)K * part * sizeof(double));
    int32_t* recA = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));
    int32_t* recB = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));
    double* lw = (double*)malloc((size_t)R * K * sizeof(double));
    double* ll = (double*)malloc((size_t)R * K * sizeof(double));
    double* bls = (double*)malloc((size_t)R * K * sizeof(double));
    double* brs = (double*)malloc((size_t)R * K * sizeof(double));
    double* ll_tilde = (double*)malloc((size_t)K * sizeof(double));
    uint64_t* cdf = (uint64_t*)malloc((size_t)K * sizeof(uint64_t));
    double* lse = (double*)malloc((size_t)R * sizeof(double));
    int32_t* lidA = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));     /* leaf id of every root slot, -1 = internal */
    int32_t* lidB = (int32_t*)malloc((size_t)K * N * sizeof(int32_t));
    uint8_t* codes = (uint8_t*)malloc((size_t)N * S);
    uint32_t* hist = (uint32_t*)calloc((size_t)N * N * 25, sizeof(uint32_t));
    if (!coreA || !coreB || !recA || !recB || !lw || !ll || !bls || !brs || !ll_tilde || !cdf || !lse || !lidA || !lidB ||
        !codes || !hist) return -1;
    const int coded = leaf_codes(genome, (size_t)N * S, codes);
    if (coded)
        for (int a = 0; a < N; ++a)
            for (int b = 0; b < N; ++b)
                for (int s = 0; s < S; ++s) ++hist[((size_t)a * N + b) * 25 + codes[(size_t)a * S + s] * 5 + codes[(size_t)b * S + s]];
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        memcpy(coreA + (size_t)k * part, genome, part * sizeof(double));
        for (int i = 0; i < N; ++i) { recA[(size_t)k * N + i] = 1; lidA[(size_t)k * N + i] = i; }
    }
    const double ll_tilde0 = ora_log(1.0 / (double)K);
    for (int k = 0; k < K; ++k) ll_tilde[k] = ll_tilde0;
    int oom = 0;
    for (int r = 0; r < R; ++r) {
        const int n = N - r, J = (n * (n - 1) / 2) * M;
        if (r > 0) {
            lse[r - 1] = weights_prepare(lw + (size_t)(r - 1) * K, K, cdf);
#pragma omp parallel for schedule(static)
            for (int k = 0; k < K; ++k) {
                uint32_t x[4];
                ora_philox((uint32_t)k, (uint32_t)r, 2u, 0u, seed, x);
                uint64_t Rr = ((uint64_t)x[1] << 32) | x[0];
                int a = cdf_search(cdf, K, ora_mulhi64(Rr, cdf[K - 1]));
                memcpy(coreB + (size_t)k * part, coreA + (size_t)a * part, (size_t)n * node * sizeof(double));
                memcpy(recB + (size_t)k * N, recA + (size_t)a * N, (size_t)n * sizeof(int32_t));
                memcpy(lidB + (size_t)k * N, lidA + (size_t)a * N, (size_t)n * sizeof(int32_t));
                ll_tilde[k] = ll[(size_t)(r - 1) * K + a];
                if (ancestors) ancestors[(size_t)(r - 1) * K + k] = a;
            }
            double* tc = coreA; coreA = coreB; coreB = tc;
            int32_t* tr_ = recA; recA = recB; recB = tr_;
            tr_ = lidA; lidA = lidB; lidB = tr_;
        }
        const double laml = lam_l[r], lamr = lam_r[r];
        const double loglaml = ora_log(laml), loglamr = ora_log(lamr);
#pragma omp parallel for schedule(dynamic, 1)
        for (int k = 0; k < K; ++k) {
            const double* src = coreA + (size_t)k * part;
            double* dst = coreB + (size_t)k * part;
            const int32_t* rs = recA + (size_t)k * N;
            int32_t* rd = recB + (size_t)k * N;
            const int32_t* ls = lidA + (size_t)k * N;
            int32_t* ld = lidB + (size_t)k * N;
            double* pot = (double*)malloc((size_t)J * sizeof(double));
            double* wv = (double*)malloc((size_t)J * sizeof(double));
            double* tmp = (double*)malloc(node * sizeof(double));
            double rowll[1024];
            if (!pot || !wv || !tmp) { oom = 1; free(pot); free(wv); free(tmp); continue; }
            for (int x = 0; x < n; ++x) rowll[x] = row_loglik(pi, src + (size_t)x * node, S);
            /* compute_potentials, vncsmc.py:379-416 (pairs r1 < r2 lexicographic, M sub-samples each) */
            int t = 0;
            for (int r1 = 0; r1 < n - 1; ++r1)
                for (int r2 = r1 + 1; r2 < n; ++r2, ++t)
                    for (int m = 0; m < M; ++m) {
                        const int j = t * M + m;
                        uint32_t x[4];
                        ora_philox((uint32_t)k, (uint32_t)r, 3u, (uint32_t)j, seed, x);
                        const double tl = (-ora_log(ora_unit_oc(x[0], x[1]))) / laml;
                        const double tr = (-ora_log(ora_unit_oc(x[2], x[3]))) / lamr;
                        double Pl[16], Pr[16];
                        if (jc) { ora_jc69(tl, Pl); ora_jc69(tr, Pr); } else { ora_expm4(Q, tl, Pl); ora_expm4(Q, tr, Pr); }
                        const double* L = src + (size_t)r1 * node;
                        const double* Rr = src + (size_t)r2 * node;
                        double merged_ll;
                        if (coded && ls[r1] >= 0 && ls[r2] >= 0) {          /* leaf-leaf pair: 25 code pairs */
                            const uint32_t* h = hist + ((size_t)ls[r1] * N + ls[r2]) * 25;
                            ora_canon64 cs;
                            ora_canon64_init(&cs);
                            for (int cp = 0; cp < 25; ++cp) {
                                if (!h[cp]) continue;
                                double Lv[4], Rv[4], o[4];
                                code_row(cp / 5, Lv);
                                code_row(cp % 5, Rv);
                                merge_site(Lv, Rv, Pl, Pr, o);
                                ora_canon64_add(&cs, cp, (double)h[cp] * ora_log(site_lik(pi, o)));
                            }
                            merged_ll = ora_canon64_total(&cs);
                        } else if (coded && ((ls[r1] >= 0) != (ls[r2] >= 0))) {
                            /* coded leaf x internal root (contract v4): the site likelihood depends on the leaf only through
                             * its code c, lik[s] = X[s] . v_c with v_c[i] = sum_j P_int[i][j] (pi_j (leaf_c . P_leaf)[j]) */
                            const int leaf_left = ls[r1] >= 0;
                            const double* Pleaf = leaf_left ? Pl : Pr;
                            const double* Pint = leaf_left ? Pr : Pl;
                            const double* X = leaf_left ? Rr : L;
                            const uint8_t* cd = codes + (size_t)(leaf_left ? ls[r1] : ls[r2]) * S;
                            double v[5][4];
                            for (int c = 0; c < 5; ++c) {
                                double u[4];
                                for (int jj = 0; jj < 4; ++jj) {
                                    const double t = c < 4 ? Pleaf[c * 4 + jj]
                                                           : ora_fma(1.0, Pleaf[12 + jj], ora_fma(1.0, Pleaf[8 + jj], ora_fma(1.0, Pleaf[4 + jj], 1.0 * Pleaf[jj])));
                                    u[jj] = pi[jj] * t;
                                }
                                for (int i = 0; i < 4; ++i) {
                                    double acc = Pint[i * 4] * u[0];
                                    acc = ora_fma(Pint[i * 4 + 1], u[1], acc);
                                    acc = ora_fma(Pint[i * 4 + 2], u[2], acc);
                                    v[c][i] = ora_fma(Pint[i * 4 + 3], u[3], acc);
                                }
                            }
                            ora_canon_lp cl;
                            ora_canon_lp_init(&cl, S);
                            for (int s = 0; s < S; ++s) {
                                const double* x = X + (size_t)s * 4;
                                const double* vc = v[cd[s]];
                                double lik = x[0] * vc[0];
                                lik = ora_fma(x[1], vc[1], lik);
                                lik = ora_fma(x[2], vc[2], lik);
                                lik = ora_fma(x[3], vc[3], lik);
                                ora_canon_lp_mul(&cl, s, lik);
                            }
                            merged_ll = ora_canon_lp_total(&cl);
                        } else {
                            for (int s = 0; s < S; ++s) merge_site(L + (size_t)s * 4, Rr + (size_t)s * 4, Pl, Pr, tmp + (size_t)s * 4);
                            merged_ll = row_loglik(pi, tmp, S);
                        }
                        double jp = merged_ll + (-log_double_factorial_count(rs[r1] + rs[r2]));
                        jp = jp - (rowll[r1] + (-log_double_factorial_count(rs[r1])));
                        jp = jp - (rowll[r2] + (-log_double_factorial_count(rs[r2])));
                        pot[j] = jp;
                    }
            if (potentials_out) {
                double* po = potentials_out + ((size_t)r * K + k) * Jmax;
                for (int j = 0; j < Jmax; ++j) po[j] = j < J ? pot[j] : 0.0;
            }
            /* normalise and draw one (pair, sub-sample), vncsmc.py:298-301, 407 */
            double mx = -ORA_INF;
            for (int j = 0; j < J; ++j) if (!ora_isnan(pot[j]) && pot[j] > mx) mx = pot[j];
            const int all_bad = !(mx > -ORA_INF) || mx == ORA_INF;
            double ssum = 0.0;
            uint64_t run = 0, thr;
            for (int j = 0; j < J; ++j) {
                wv[j] = all_bad ? 1.0 : (ora_isnan(pot[j]) ? 0.0 : ora_exp(pot[j] - mx));
                ssum = ssum + wv[j];
            }
            uint64_t total = 0;
            for (int j = 0; j < J; ++j) total += all_bad ? 1ull : (uint64_t)(wv[j] * 17592186044416.0);
            {
                uint32_t x[4];
                ora_philox((uint32_t)k, (uint32_t)r, 3u, 0xFFFFFFFFu, seed, x);
                thr = ora_mulhi64(((uint64_t)x[1] << 32) | x[0], total);
            }
            int jsel = J - 1;
            for (int j = 0; j < J; ++j) {
                run += all_bad ? 1ull : (uint64_t)(wv[j] * 17592186044416.0);
                if (run > thr) { jsel = j; break; }
            }
            const double logq = pot[jsel] - ((all_bad ? 0.0 : mx) + ora_log(ssum));
            /* chosen pair and its sub-sample's branches (vncsmc.py:301, 315-320) */
            int il = 0, ir = 1;
            t = 0;
            for (int r1 = 0; r1 < n - 1; ++r1)
                for (int r2 = r1 + 1; r2 < n; ++r2, ++t)
                    if (t == jsel / M) { il = r1; ir = r2; }
            uint32_t x[4];
            ora_philox((uint32_t)k, (uint32_t)r, 3u, (uint32_t)jsel, seed, x);
            const double tl = (-ora_log(ora_unit_oc(x[0], x[1]))) / laml;
            const double tr = (-ora_log(ora_unit_oc(x[2], x[3]))) / lamr;
            bls[(size_t)r * K + k] = tl;
            brs[(size_t)r * K + k] = tr;
            double Pl[16], Pr[16];
            if (jc) { ora_jc69(tl, Pl); ora_jc69(tr, Pr); } else { ora_expm4(Q, tl, Pl); ora_expm4(Q, tr, Pr); }
            int nrem = 0;
            for (int i = n - 1; i >= 0; --i) {                     /* remaining slots, descending (vncsmc.py:305) */
                if (i == il || i == ir) continue;
                memcpy(dst + (size_t)nrem * node, src + (size_t)i * node, node * sizeof(double));
                rd[nrem] = rs[i];
                ld[nrem] = ls[i];
                ++nrem;
            }
            ld[nrem] = -1;
            double* nw = dst + (size_t)nrem * node;
            for (int s = 0; s < S; ++s) merge_site(src + (size_t)il * node + (size_t)s * 4, src + (size_t)ir * node + (size_t)s * 4, Pl, Pr, nw + (size_t)s * 4);
            rd[nrem] = rs[il] + rs[ir];
            if (nodes_out) memcpy(nodes_out + ((size_t)r * K + k) * node, nw, node * sizeof(double));
            if (merges) { merges[((size_t)r * K + k) * 2] = il; merges[((size_t)r * K + k) * 2 + 1] = ir; }
            double fl = 0.0, fp = 0.0;
            int vminus = 0;
            for (int xr = 0; xr < n - 1; ++xr) {
                fl = fl + row_loglik(pi, dst + (size_t)xr * node, S);
                fp = fp + (-log_double_factorial_count(rd[xr]));
                vminus += rd[xr] - (rd[xr] == 1 ? 1 : 0);
            }
            double lp = 0.0, rp = 0.0;
            for (int j = 0; j <= r; ++j) {
                lp = lp + ((-laml) * bls[(size_t)j * K + k] + loglaml);
                rp = rp + ((-lamr) * brs[(size_t)j * K + k] + loglamr);
            }
            const double llr = ((fl + fp) + lp) + rp;
            const double paren = ((loglaml - laml * tl) + loglamr) - lamr * tr;
            ll[(size_t)r * K + k] = llr;
            lw[(size_t)r * K + k] = (((llr - ll_tilde[k]) - paren) + ora_log((double)vminus)) - logq;   /* vncsmc.py:489-491 */
            free(pot); free(wv); free(tmp);
        }
        { double* tc = coreA; coreA = coreB; coreB = tc; int32_t* tr_ = recA; recA = recB; recB = tr_;
          tr_ = lidA; lidA = lidB; lidB = tr_; }
    }
    lse[R - 1] = weights_prepare(lw + (size_t)(R - 1) * K, K, NULL);
    free(lidA); free(lidB); free(codes); free(hist);
    double z = 0.0;
    for (int r = 0; r < R; ++r) z = z + lse[r];
    if (logZ) *logZ = z;
    if (log_weights) memcpy(log_weights, lw, (size_t)R * K * sizeof(double));
    if (log_lik) memcpy(log_lik, ll, (size_t)R * K * sizeof(double));
    if (lbranch) memcpy(lbranch, bls, (size_t)R * K * sizeof(double));
    if (rbranch) memcpy(rbranch, brs, (size_t)R * K * sizeof(double));
    free(coreA); free(coreB); free(recA); free(recB); free(lw); free(ll); free(bls); free(brs);
    free(ll_tilde); free(cdf); free(lse);
    return oom ? -1 : 0;
}
