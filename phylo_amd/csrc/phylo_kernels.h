// phylo_kernels.h -- HIP kernels of the CSMC hot path for gfx950 (MI355X).  See DESIGN.md for the
// data layout and the per-kernel roofline accounting.  All arithmetic follows phylo_math.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phylo_math.h"

#define PK_COLS 256          // canonical site-sum columns == threads per merge workgroup
#define PK_AUX 8             // per-particle scalars handed from the bookkeeping kernel to the merge epilogue
#define PK_MAX_TAXA 512
#define PK_MAX_GROUPS 64                 // independent sweeps batched in one context

// aux slots
enum { AUX_SUM_REM = 0, AUX_FPRIOR, AUX_LPRIOR, AUX_RPRIOR, AUX_LL_TILDE, AUX_PAREN, AUX_LOGV, AUX_Q };

// ------------------------------------------------------------------------------------------------
// Canonical sum over 256 columns of a workgroup (the sum over PARTICLES of the resampling weights: element k
// belongs to column k mod 256): adjacent-pair tree inside each 64-lane wave (xor butterfly 1,2,...,32:
// a+b == b+a bitwise, so every lane ends with the same value), then the four wave totals left to right.
// oracle/csrc/oracle.c mirrors exactly this tree.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pk_block_canon_sum(double col, double* sh4) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) col = col + __shfl_xor(col, off, 64);
    __syncthreads();   // protect sh4 from a previous use
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = col;
    __syncthreads();
    return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}

// ------------------------------------------------------------------------------------------------
// Canonical sum over SITES (contract v5, phylo_math.h): one wave owns a (row, tile); lane l holds the finished column l
// (the log of the running product of sites tile start + l + 64 j, pm_lp); the 64 columns are added by the adjacent-pair
// tree below, tile values left to right.  Levels 1..8 are DPP moves inside a row of 16 lanes (after levels 1 and 2 a quad
// holds one value, so the mirrored partner of level 4 / 8 holds exactly the value of the xor partner); the four row totals
// are read by lane and added as (r0 + r1) + (r2 + r3).  a + b is commutative bit for bit: the xor butterfly's result without
// six dependent trips through the LDS crossbar.  Wave-uniform result.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double pk_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pk_readlane(double v, int l) {      // l wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// Scalar-cache touch.  A wave's start-up is a chain of scalar-load round trips (kernel arguments -> ids -> matrices -> pi -> first
// rows), and hipcc requests a wave-uniform matrix only where it is first used, behind the branches on the ids (requested earlier
// its 32-64 scalar registers would live across those branches: spills).  PK_TOUCH_* issue one-dword scalar loads of the
// matrix's cache lines BESIDE the load of the ids, so the real loads hit the scalar cache: headline +4.4 %, single sweep -1 %
// (A/B/A/B on one box).  The dummy destination registers must not get a new owner while a touch is in flight, on ANY path:
// PK_TOUCH_END(ids) waits for scalar memory and passes the ids THROUGH (an in-out operand), so it sits before every use of the
// ids -- i.e. on every path, before the first branch on them -- and reads the dummies, which keeps them reserved from the touch
// to the wait.  Neither asm is volatile: a volatile asm counts as a store, after which hipcc no longer proves uniform loads
// invariant and turns the matrices' scalar loads into vector loads (32 more VGPRs, scratch).  (First version: a consumer after
// the matrices' loads and without the wait -- on the paths that did not reach it hipcc gave the dummies' registers to its own
// loads while the touches were in flight, and the twisted sweep faulted.)
__device__ __forceinline__ const char* pk_uniform_ptr(const void* p);
#ifndef PK_NO_SCALAR_TOUCH
#define PK_TOUCH_DECL unsigned int pk_t0_, pk_t1_, pk_t2_, pk_t3_, pk_t4_, pk_t5_
#define PK_TOUCH_256_2(p256, q, r)                                                                                              \
    asm("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\ts_load_dword %3, %6, 0xc0\n\t"      \
        "s_load_dword %4, %7, 0x0\n\ts_load_dword %5, %8, 0x0"                                                                 \
        : "=&s"(pk_t0_), "=&s"(pk_t1_), "=&s"(pk_t2_), "=&s"(pk_t3_), "=&s"(pk_t4_), "=&s"(pk_t5_)                             \
        : "s"(pk_uniform_ptr(p256)), "s"(pk_uniform_ptr(q)), "s"(pk_uniform_ptr(r)))   /* (readfirstlane: a uniform value may live in VGPRs) */
#define PK_TOUCH_END(x) asm("s_waitcnt lgkmcnt(0)" : "+s"(x) : "s"(pk_t0_), "s"(pk_t1_), "s"(pk_t2_), "s"(pk_t3_), "s"(pk_t4_), "s"(pk_t5_))
#else
#define PK_TOUCH_DECL
#define PK_TOUCH_256_2(p256, q, r)
#define PK_TOUCH_END(x)
#endif

__device__ __forceinline__ double pk_wave_tree_sum(double v) {
    v = v + pk_dpp<0xB1>(v);          // quad_perm [1,0,3,2]: xor 1
    v = v + pk_dpp<0x4E>(v);          // quad_perm [2,3,0,1]: xor 2
    v = v + pk_dpp<0x141>(v);         // row_half_mirror: the other quad of my 8
    v = v + pk_dpp<0x140>(v);         // row_mirror: the other 8 of my row
    const double r0 = pk_readlane(v, 0), r1 = pk_readlane(v, 16), r2 = pk_readlane(v, 32), r3 = pk_readlane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// Partial-likelihood rows always live in global memory (leaves, this rank's pool or a peer's mapped pool), but
// a pool base fetched from the pointer table is a generic pointer to the compiler, which then emits flat_load
// (uncounted waits, no software pipelining).  Loads go through an explicit global-address-space pointer.
typedef double pk_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const pk_d2 pk_gd2c;
typedef __attribute__((address_space(1))) pk_d2 pk_gd2;
// LDS written by some lanes of a wave and read by others of the SAME wave: the LDS pipe serves a wave's instructions in
// order, so only the compiler has to be kept from moving the reads up
__device__ __forceinline__ void pk_wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ pk_d2 pk_gload2(const double* p) { return *(pk_gd2c*)p; }

__device__ __forceinline__ void pk_load4(const double* __restrict__ p, double* v) {
    const pk_d2 a = pk_gload2(p), b = pk_gload2(p + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

__device__ __forceinline__ void pk_store4(double* __restrict__ p, const double* v) {
    reinterpret_cast<double2*>(p)[0] = make_double2(v[0], v[1]);
    reinterpret_cast<double2*>(p)[1] = make_double2(v[2], v[3]);
}

// streaming (non-temporal) form: the new node is read again only by the few particles that survive the next
// resampling, so it should not displace the leaves and live ancestors from L2
__device__ __forceinline__ void pk_store4_nt(double* __restrict__ p, const double* v) {
    pk_d2 a = {v[0], v[1]}, b = {v[2], v[3]};
    __builtin_nontemporal_store(a, reinterpret_cast<pk_d2*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<pk_d2*>(p) + 1);
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
__global__ __launch_bounds__(64) void pk_expm_batched(const double* __restrict__ Q, const double* __restrict__ t, int n, int jc,
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

// all branch lengths and transition matrices of one sweep: thread per (rank event r, local particle k,
// side).  b = -log(U)/lambda_r (vcsmc.py:351-356); Pmat[r][k] = {P(b_l), P(b_r)}.
__device__ __forceinline__ void pk_sweep_draws_body(int t, const double* __restrict__ Q, const double* __restrict__ lam_l,
                                                    const double* __restrict__ lam_r, int jc, uint64_t seed, int R, int K,
                                                    int k0, double* __restrict__ bl, double* __restrict__ br,
                                                    double* __restrict__ Pmat, int Kg, const uint64_t* __restrict__ group_seeds) {
    if (t >= 2 * R * K) return;
    const int side = t & 1, i = t >> 1;
    const int r = i / K, k = i - r * K;
    int kp = k0 + k;                                      // particle index of the RNG contract
    if (group_seeds) { const int g = kp / Kg; seed = group_seeds[g]; kp -= g * Kg; }
    const pm_u32x4 x = pm_philox4x32((uint32_t)kp, (uint32_t)r, PM_STREAM_BRANCH, 0u, seed);
    const double b = side ? (-pm_log(pm_unit_oc(x.z, x.w))) / lam_r[r] : (-pm_log(pm_unit_oc(x.x, x.y))) / lam_l[r];
    (side ? br : bl)[i] = b;
    double q[16], p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = Q[j];
    if (jc) pm_jc69(b, p); else pm_expm4(q, b, p);
    double* out = Pmat + (size_t)i * 32 + side * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) out[j] = p[j];
}

// The prologue of a sweep of the plain proposal as ONE launch: the draws above, the initial root tables (pk_init_tables) and the
// cleared marks of the lazy nodes (was a hipMemsetAsync, i.e. a fill kernel) are independent of one another.
struct pk_prologue_args {
    const double *Q, *lam_l, *lam_r;
    int jc, R, Kloc, k0, Kg;
    uint64_t seed;
    double *bl, *br, *Pmat;
    const uint64_t* group_seeds;
    unsigned long long* rdraw;          // [R][K] or NULL: the 64-bit resampling draw of every GLOBAL particle at every rank event
    int32_t *roots, *cnt;
    double* rootll;
    const double* nodell;
    int K, N;
    unsigned int* mark;                 // NULL: no marks to clear
    unsigned int mark_words;            // multiple of 4
    int draw_blocks, init_blocks, mark_blocks;   // then the blocks that fill rdraw
};
// The draws of a LARGE launch (batched sweeps), sorted by the work they need.  pm_expm4 picks its Pade order per matrix from
// ||Q b||_1; with one matrix per lane in index order nearly every wave holds all five orders and runs all five branches
// (~2500 VALU instructions per wave, 70 % of the lanes needing order 5 only).  Here a workgroup of 256 threads takes
// PK_DRAW_ITEMS matrices: pass 1 draws the branch lengths (one Philox evaluation serves both sides of a particle) and files
// every matrix under an ESTIMATE of its order (b ||Q||_1 against the thresholds) in LDS; pass 2 walks the list so that a wave's
// 64 matrices are of one class except where two classes meet.  pm_expm4 itself is unchanged and decides the order as before:
// the estimate only schedules, it cannot change a bit.
#define PK_DRAW_ITEMS 1024
__device__ __forceinline__ void pk_sweep_draws_sorted(int wg, const pk_prologue_args& p) {
    __shared__ double sb[PK_DRAW_ITEMS];
    __shared__ unsigned short slist[PK_DRAW_ITEMS];
    __shared__ int scount[8];
    const int tid = threadIdx.x, NT = 256;
    const int total = 2 * p.R * p.Kloc, base = wg * PK_DRAW_ITEMS;
    if (tid < 8) scount[tid] = 0;
    double q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = p.Q[j];
    double qn = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cs = cs + (q[i * 4 + j] < 0.0 ? -q[i * 4 + j] : q[i * 4 + j]);
        qn = cs > qn ? cs : qn;
    }
    __syncthreads();
    int cls[PK_DRAW_ITEMS / 512][2], pos[PK_DRAW_ITEMS / 512][2];
#pragma unroll
    for (int u = 0; u < PK_DRAW_ITEMS / 512; ++u) {
        const int pair = u * NT + tid, t0 = base + 2 * pair;        // (base is even: both sides of a particle in one workgroup)
        cls[u][0] = cls[u][1] = -1;
        if (t0 < total) {
            const int i = t0 >> 1, r = i / p.Kloc, k = i - r * p.Kloc;
            int kp = p.k0 + k;
            uint64_t seed = p.seed;
            if (p.group_seeds) { const int g = kp / p.Kg; seed = p.group_seeds[g]; kp -= g * p.Kg; }
            const pm_u32x4 x = pm_philox4x32((uint32_t)kp, (uint32_t)r, PM_STREAM_BRANCH, 0u, seed);
            const double b0 = (-pm_log(pm_unit_oc(x.x, x.y))) / p.lam_l[r];
            const double b1 = (-pm_log(pm_unit_oc(x.z, x.w))) / p.lam_r[r];
            p.bl[i] = b0;
            p.br[i] = b1;
            sb[2 * pair] = b0;
            sb[2 * pair + 1] = b1;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const double nrm = (sd ? b1 : b0) * qn;
                const int c = (nrm > 1.495585217958292e-2) + (nrm > 2.539398330063230e-1) + (nrm > 9.504178996162932e-1) +
                              (nrm > 2.097847961257068e0);
                cls[u][sd] = c;
                pos[u][sd] = atomicAdd(&scount[c], 1);
            }
        }
    }
    __syncthreads();
    int off[5];
    off[0] = 0;
#pragma unroll
    for (int c = 1; c < 5; ++c) off[c] = off[c - 1] + scount[c - 1];
#pragma unroll
    for (int u = 0; u < PK_DRAW_ITEMS / 512; ++u)
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
            if (cls[u][sd] >= 0) {
                const int c = cls[u][sd];
                const int o = c == 0 ? off[0] : c == 1 ? off[1] : c == 2 ? off[2] : c == 3 ? off[3] : off[4];
                slist[o + pos[u][sd]] = (unsigned short)(2 * (u * NT + tid) + sd);
            }
    __syncthreads();
    // (a matrix is one 128-byte line of Pmat; eight lanes writing one line, 16 bytes each, through LDS: measured, no gain)
    const int nitems = total - base < PK_DRAW_ITEMS ? total - base : PK_DRAW_ITEMS;
    #pragma unroll 1
    for (int sidx = tid; sidx < nitems; sidx += NT) {
        const int li = slist[sidx], t = base + li;
        double pm[16];
        pm_expm4(q, sb[li], pm);
        double* out = p.Pmat + (size_t)(t >> 1) * 32 + (t & 1) * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = pm[j];
    }
}

template <int NT, bool SORTED>
__device__ __forceinline__ void pk_sweep_prologue_body(const pk_prologue_args& p) {
    const int b = blockIdx.x;
    if (b < p.draw_blocks) {
        if constexpr (SORTED) pk_sweep_draws_sorted(b, p);
        else pk_sweep_draws_body(b * NT + (int)threadIdx.x, p.Q, p.lam_l, p.lam_r, p.jc, p.seed, p.R, p.Kloc, p.k0, p.bl, p.br, p.Pmat, p.Kg,
                                 p.group_seeds);
    } else if (b < p.draw_blocks + p.init_blocks) {
        const int i0 = ((b - p.draw_blocks) * NT + (int)threadIdx.x) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < p.K * p.N) {
                p.roots[i] = i % p.N;
                p.cnt[i] = 1;
                p.rootll[i] = p.nodell[i % p.N];
            }
        }
    } else if (b < p.draw_blocks + p.init_blocks + p.mark_blocks) {
        const unsigned int w = (unsigned int)((b - p.draw_blocks - p.init_blocks) * NT + (int)threadIdx.x) * 4u;
        if (w < p.mark_words) *reinterpret_cast<uint4*>(p.mark + w) = make_uint4(0u, 0u, 0u, 0u);
    } else {                                              // the draw the index search of rank event r >= 1 scales by the cdf total
        const long i = (long)(b - p.draw_blocks - p.init_blocks - p.mark_blocks) * NT + (int)threadIdx.x + p.K;   // rows 1 .. R-1
        if (i < (long)p.R * p.K) {
            const int r = (int)(i / p.K);
            int kp = (int)(i - (long)r * p.K);
            uint64_t seed = p.seed;
            if (p.group_seeds) { const int g = kp / p.Kg; seed = p.group_seeds[g]; kp -= g * p.Kg; }
            const pm_u32x4 d = pm_philox4x32((uint32_t)kp, (uint32_t)r, PM_STREAM_RESAMPLE, 0u, seed);
            p.rdraw[i] = ((unsigned long long)d.y << 32) | d.x;
        }
    }
}
__global__ __launch_bounds__(64) void pk_sweep_prologue(const pk_prologue_args p) { pk_sweep_prologue_body<64, false>(p); }
__global__ __launch_bounds__(256) void pk_sweep_prologue_sorted(const pk_prologue_args p) { pk_sweep_prologue_body<256, true>(p); }

// ------------------------------------------------------------------------------------------------
// canonical sum_s log(pi . x[s]) of `rows` vectors [S,4]; one wave per row, its tiles (T sites) one after the other.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pk_row_loglik(const double* __restrict__ x, const double* __restrict__ pi4,
                                                     int S, int T, double* __restrict__ out) {
    const double* row = x + (size_t)blockIdx.x * S * 4;
    double pi[4] = {pi4[0], pi4[1], pi4[2], pi4[3]};
    double tot = 0.0;
    for (int s0 = 0; s0 < S; s0 += T) {
        const int s1 = s0 + T < S ? s0 + T : S;
        pm_lp col = pm_lp_init();
        for (int s = s0 + threadIdx.x; s < s1; s += 64) {
            double v[4];
            pk_load4(row + (size_t)s * 4, v);
            pm_lp_mul(col, pk_site_lik(pi, v));
        }
        const double t = pk_wave_tree_sum(pm_lp_finish(col));
        tot = s0 ? tot + t : t;
    }
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
// k5: resampling.  One 256-thread workgroup: max, canonical fp sum of the weights (log-normaliser),
// integer weights floor(exp(logw - max) 2^44) and their inclusive prefix sum -> cdf[K] (uint64).
// lse_out = logsumexp_k(logw) - log K.  Loads are agent-scope (sc1) so the same code can run as the tail
// of the merge kernel on weights other workgroups of the same launch have just published.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pk_ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pk_st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct pk_scan_lds {
    double d4[4];
    unsigned long long u4[4];
};

// eight agent-scope (sc1) 8-byte loads issued back to back, ONE wait: hipcc waits after every relaxed
// atomic load it emits itself, which serialises the round trips (cdna_hip_programming.md 5.7, form (i)).
__device__ __forceinline__ void pk_ld8_agent(const double* const (&p)[8], double (&v)[8]) {
    asm volatile(
        "global_load_dwordx2 %0, %8, off sc1\n\t"
        "global_load_dwordx2 %1, %9, off sc1\n\t"
        "global_load_dwordx2 %2, %10, off sc1\n\t"
        "global_load_dwordx2 %3, %11, off sc1\n\t"
        "global_load_dwordx2 %4, %12, off sc1\n\t"
        "global_load_dwordx2 %5, %13, off sc1\n\t"
        "global_load_dwordx2 %6, %14, off sc1\n\t"
        "global_load_dwordx2 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}

// load the tile [base, base + 2048) of logw, element base + tid + 256 j -> v[j]; out of range -> NaN
__device__ __forceinline__ void pk_scan_tile(const double* logw, int K, int base, double (&v)[8]) {
    const double* p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = base + (int)threadIdx.x + PK_COLS * j;
        p[j] = logw + (k < K ? k : K - 1);
    }
    pk_ld8_agent(p, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (base + (int)threadIdx.x + PK_COLS * j >= K) v[j] = pm_nan();
}

// `stage`: K-element scratch for the integer weights between the two passes: LDS when it fits, else cdf[].
// `publish`: the final cdf is written with agent-scope (write-through) 8-byte stores, because workgroups
// of the SAME launch read it after a flag hand-off (pk_rank_scan_book).
__device__ __forceinline__ void pk_scan_block(const double* logw, int K, uint64_t* __restrict__ cdf,
                                              double* __restrict__ lse_out, pk_scan_lds* sh,
                                              unsigned long long* stage, bool publish = false) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool one_tile = K <= 8 * PK_COLS;
    double v[8];
    // ---- max (NaN counts as -inf)
    double m = -pm_inf();
    for (int base = 0; base < K; base += 8 * PK_COLS) {
        pk_scan_tile(logw, K, base, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!pm_isnan(v[j]) && v[j] > m) m = v[j];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    __syncthreads();
    if (lane == 0) sh->d4[wv] = m;
    __syncthreads();
    m = sh->d4[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) m = sh->d4[i] > m ? sh->d4[i] : m;
    const bool all_bad = !(m > -pm_inf()) || m == pm_inf();
    // ---- canonical fp sum (thread t = column t: elements t, t+256, ... in increasing order) and the
    //      integer weights, staged in cdf[] itself
    double col = 0.0;
    for (int base = 0; base < K; base += 8 * PK_COLS) {
        if (!one_tile) pk_scan_tile(logw, K, base, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = base + tid + PK_COLS * j;
            if (k < K) {
                const double w = all_bad ? 1.0 : (pm_isnan(v[j]) ? 0.0 : pm_exp(v[j] - m));
                col = col + w;
                if (cdf) stage[k] = all_bad ? 1ull : (unsigned long long)(w * PM_CDF_SCALE);
            }
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) col = col + __shfl_xor(col, off, 64);
    __syncthreads();
    if (lane == 0) sh->d4[wv] = col;
    __syncthreads();                                   // also orders the staging stores (workgroup scope)
    if (tid == 0 && lse_out) {
        const double sum = ((sh->d4[0] + sh->d4[1]) + sh->d4[2]) + sh->d4[3];
        const double mm = all_bad ? 0.0 : m;
        *lse_out = (mm + pm_log(sum)) - pm_log((double)K);
    }
    if (!cdf) return;
    // ---- integer inclusive scan in place, tile by tile; inside a tile thread t owns the 8 contiguous
    //      elements [base + 8t, base + 8t + 8) (independent 16-byte loads, one round trip)
    unsigned long long carry = 0;
    for (int base = 0; base < K; base += 8 * PK_COLS) {
        const int lo = base + 8 * tid;
        unsigned long long e[8];
        if (lo + 8 <= K) {
            const ulonglong2* q = reinterpret_cast<const ulonglong2*>(stage + lo);
            const ulonglong2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            e[0] = q0.x; e[1] = q0.y; e[2] = q1.x; e[3] = q1.y; e[4] = q2.x; e[5] = q2.y; e[6] = q3.x; e[7] = q3.y;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (lo + j < K) ? stage[lo + j] : 0ull;
        }
#pragma unroll
        for (int j = 1; j < 8; ++j) e[j] += e[j - 1];
        const unsigned long long local = e[7];
        unsigned long long incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        __syncthreads();
        if (lane == 63) sh->u4[wv] = incl;
        __syncthreads();
        unsigned long long run = carry + incl - local;
        for (int i = 0; i < wv; ++i) run += sh->u4[i];
        if (publish) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (lo + j < K)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(cdf) + lo + j, run + e[j], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        } else if (lo + 8 <= K) {
            ulonglong2* q = reinterpret_cast<ulonglong2*>(cdf + lo);
            q[0] = make_ulonglong2(run + e[0], run + e[1]);
            q[1] = make_ulonglong2(run + e[2], run + e[3]);
            q[2] = make_ulonglong2(run + e[4], run + e[5]);
            q[3] = make_ulonglong2(run + e[6], run + e[7]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (lo + j < K) cdf[lo + j] = run + e[j];
        }
        carry += ((sh->u4[0] + sh->u4[1]) + sh->u4[2]) + sh->u4[3];
    }
}

#define PK_SCAN_LDS_MAX_K 8192
__global__ __launch_bounds__(PK_COLS) void pk_resample_scan(const double* logw, int K, uint64_t* __restrict__ cdf,
                                                             double* __restrict__ lse_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [pk_scan_lds][K x u64 when K fits]
    pk_scan_lds* sh = reinterpret_cast<pk_scan_lds*>(smem);
    unsigned long long* stage = (K <= PK_SCAN_LDS_MAX_K) ? reinterpret_cast<unsigned long long*>(smem + sizeof(pk_scan_lds))
                                                         : reinterpret_cast<unsigned long long*>(cdf);
    pk_scan_block(logw, K, cdf, lse_out, sh, stage);
}
// G independent sweeps batched in one context: one workgroup per group scans its own K-segment
__global__ __launch_bounds__(PK_COLS) void pk_resample_scan_groups(const double* logw, int K, uint64_t* __restrict__ cdf,
                                                                    double* __restrict__ lse_out, int lse_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pk_scan_lds* sh = reinterpret_cast<pk_scan_lds*>(smem);
    const int g = blockIdx.x;
    uint64_t* cg = cdf ? cdf + (size_t)g * K : nullptr;
    unsigned long long* stage = (K <= PK_SCAN_LDS_MAX_K) ? reinterpret_cast<unsigned long long*>(smem + sizeof(pk_scan_lds))
                                                         : reinterpret_cast<unsigned long long*>(cg);
    pk_scan_block(logw + (size_t)g * K, K, cg, lse_out + (size_t)g * lse_stride, sh, stage);
}
__host__ inline size_t pk_scan_lds_bytes(int K) {
    return sizeof(pk_scan_lds) + (K <= PK_SCAN_LDS_MAX_K ? (size_t)K * 8 : 0);
}

__device__ __forceinline__ int pk_cdf_search(const uint64_t* __restrict__ cdf, int K, uint64_t thr) {
    int lo = 0, hi = K;             // first index with cdf[i] > thr
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > thr) hi = mid; else lo = mid + 1;
    }
    return lo < K ? lo : K - 1;
}

// the same search by one wave: 64 probes per step.  The first round's probe addresses do not depend on the
// threshold, so `total` and the coarse probes travel together: 2 dependent round trips for K <= 4096.
__device__ __forceinline__ uint64_t pk_cdf_ld(const uint64_t* cdf, int i, bool agent) {
    if (agent) return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(cdf) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return cdf[i];
}
__device__ __forceinline__ int pk_cdf_search_wave(const uint64_t* cdf, int K, uint64_t R, int lane, bool agent = false) {
    int lo = 0, hi = K;             // invariant: answer in [lo, hi), cdf[hi-1] > thr
    int step = (K + 63) >> 6;
    int p = (lane + 1) * step - 1;
    if (p > K - 1) p = K - 1;
    const uint64_t total = pk_cdf_ld(cdf, K - 1, agent);
    uint64_t c = pk_cdf_ld(cdf, p, agent);
    const uint64_t thr = pm_mulhi64(R, total);
    for (;;) {
        const unsigned long long mask = __ballot(c > thr);
        const int f = mask ? __ffsll((long long)mask) - 1 : 63;
        int pf = lo + (f + 1) * step - 1;
        if (pf > hi - 1) pf = hi - 1;
        lo = lo + f * step;
        hi = pf + 1;
        if (lo >= hi) lo = hi - 1;
        if (hi - lo <= 1) break;
        step = (hi - lo + 63) >> 6;
        p = lo + (lane + 1) * step - 1;
        if (p > hi - 1) p = hi - 1;
        c = pk_cdf_ld(cdf, p, agent);
    }
    return lo;
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

__global__ void pk_logz_total_groups(double* __restrict__ lse, int R, int stride) {     // lse[g][0..R-1] -> lse[g][R]
    if (threadIdx.x == 0) {
        double* l = lse + (size_t)blockIdx.x * stride;
        double z = 0.0;
        for (int r = 0; r < R; ++r) z = z + l[r];
        l[R] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// sweep state
// ------------------------------------------------------------------------------------------------
__global__ void pk_init_tables(int32_t* __restrict__ roots, int32_t* __restrict__ cnt, double* __restrict__ rootll,
                               const double* __restrict__ nodell, int K, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * N) return;
    roots[i] = i % N;
    cnt[i] = 1;
    rootll[i] = nodell[i % N];      // sum_s log(pi . leaf[s]) of the leaf in that slot
}

// Arguments of one rank event.  Integer state (root tables) is indexed by GLOBAL particle; float state of
// this rank's shard by local slot k = kg - k0.
struct pk_rank_args {
    int r, n, N, S, K /*global*/, Kloc, k0;
    uint64_t seed;
    uint32_t flags;
    const int32_t* roots_old; const int32_t* cnt_old;     // [K][N]
    int32_t* roots_new; int32_t* cnt_new;                 // [K][N]
    const double* rootll_old; double* rootll_new;         // [K][N]: sum_s log(pi . x[s]) of the root in each slot
    const uint64_t* cdf;                                  // [K] (r > 0): scan of log w_{r-1}
    const double* ll_prev;                                // ll[r-1][K] (r > 0)
    double* nodell;                                       // per node id
    const double* ldf; int ldf_n;                         // log (2 max(c,2) - 3)!! by leaf count
    const double* bl; const double* br;                   // [R][Kloc]
    double lam_l, lam_r, loglam_l, loglam_r, ll_tilde0;
    const double* leaves;                                 // [N][S][4]
    const uint8_t* leaf_codes;                            // [N][S] 0..3 one-hot state, 4 all-ones; NULL if some row is neither
    double* pool;                                         // [(N-1)][Kloc][S][4]: this rank's nodes
    const double* const* pool_ptrs;                       // [world]: every rank's pool as mapped in this process
    // sharded: remote nodes this rank has merged once are kept in a local cache (pk_pull_remote_children); mirror[node - N] is
    // 0 = not here, slot + 1, or -2 = the cache was full (read in place).  NULL: every remote child is read in place.
    int32_t* mirror;                                      // [(N-1) K], then [0] = slots taken
    double* cache;                                        // [cache_cap][S][4]
    int cache_cap;
    const double* Pmat;                                   // [Kloc][32] of this rank
    const double* pi;
    double* logw_r; double* ll_r;                         // [K] rows (global columns)
    int32_t* merges;                                      // [R][Kloc][2]
    int64_t* ancestors;                                   // [R-1][Kloc]
    // scan -> bookkeeping hand-off inside ONE launch (pk_rank_scan_book): workgroup 0 scans log w_{r-1}, publishes
    // cdf[] write-through and then stores `epoch` into *flag; the bookkeeping workgroups poll it (bounded).
    const double* scan_logw; uint64_t* scan_cdf; double* scan_lse;
    unsigned int* flag; unsigned int epoch; unsigned int* timeout_word;
    // lazy nodes (single GPU, plain proposal): the merge kernel does not store the new node; a node is written
    // only when some particle adopts its creator's table at the next resampling (pk_materialize_node)
    int lazy;
    unsigned int* mark;                                   // [R][K] 0/1: node (r, k) is in the pool
    const int32_t* child_all;                             // [R][Kloc][2] children of every node created so far
    const double* Pmat_all;                               // [R][Kloc][32]
    int32_t* child;                                       // [Kloc][2] (row r of child_all): node ids merged at this rank event
    double* aux;                                          // [Kloc][PK_AUX]: weight terms for the merge epilogue
    int32_t* pos_hist;                                    // [K][N] or NULL (PHYLO_KEEP_GRAPH): adopted slot -> new position, -1 = merged
    // sharded local bookkeeping: the grid covers this rank's particles only; an ancestor's rows of the previous
    // plane are read from its OWNER's table slab (peer mapping), byte offsets of that plane inside the slab
    const char* const* tab_ptrs;                          // [world] or NULL
    size_t tab_off_rootll, tab_off_roots, tab_off_cnt;
    // G independent sweeps batched in one context (one GPU, plain proposal): particle kg belongs to group kg / Kg,
    // draws with (kg % Kg, group_seeds[group]) and resamples inside its group's cdf segment.  Kg == K when unbatched.
    int Kg;
    const uint64_t* group_seeds;                          // [G] or NULL (use `seed`)
    int no_store;                                         // the merge does not store its node (last rank event: never read again)
    const unsigned long long* rdraw;                      // [K] 64-bit resampling draws of this rank event (pk_rank_book_mat)
    // contract v5: site tile T (multiple of 64), ntiles = ceil(S / T).  With more than one tile a merge wave leaves its tile's
    // value in tilev[k][tile] and pk_tile_epilogue adds them left to right and finishes the particle.
    int T, ntiles;
    double* tilev;                                        // [Kloc][ntiles] or NULL (ntiles == 1)
};

// LDS carve of the bookkeeping prologue (arrays of length N rounded up to a multiple of 4)
struct pk_book_lds {
    double *ord_ll, *ord_ldf, *hbl, *hbr, *anc_ll, *ldf;   // ldf: N+1 entries (n4 + 4 reserved)
    uint32_t* key; int32_t *ro, *co, *ord_cnt;
    double* aux; int32_t* misc;   // misc: [0] child l, [1] child r, [2] is_last, [3] ancestor
};
__host__ __device__ inline size_t pk_book_lds_bytes(int N) {
    const size_t n4 = ((size_t)N + 3) & ~(size_t)3;
    return (6 * n4 + 4) * 8 + PK_AUX * 8 + (4 * n4) * 4 + 16 /*misc*/;
}
__device__ __forceinline__ pk_book_lds pk_book_carve(char* base, int N) {
    const size_t n4 = ((size_t)N + 3) & ~(size_t)3;
    pk_book_lds L;
    L.ord_ll = (double*)base;
    L.ord_ldf = L.ord_ll + n4;
    L.hbl = L.ord_ldf + n4;
    L.hbr = L.hbl + n4;
    L.anc_ll = L.hbr + n4;
    L.ldf = L.anc_ll + n4;
    L.aux = L.ldf + n4 + 4;
    L.key = (uint32_t*)(L.aux + PK_AUX);
    L.ro = (int32_t*)(L.key + n4);
    L.co = L.ro + n4;
    L.ord_cnt = L.co + n4;
    L.misc = L.ord_cnt + n4;
    return L;
}

// Bookkeeping of one rank event for ONE particle (global index kg), by the first wave of a workgroup;
// every thread of the workgroup must call it (it contains workgroup barriers).
//   resampling index (vcsmc.py:285) -> adoption of the ancestor's root table (the tf.gather of :286-288, on
//   integer tables instead of partial likelihoods) -> uniform pair pick (:303-305) -> new root table
//   (:361-373) -> the scalar terms of the weight that do not depend on the new node (:376-392).
// `local`: the particle belongs to this rank's shard (float terms and outputs are produced).
// Dependent global round trips: {cdf total + coarse probes, branch history, ldf table} -> {fine probes}
// -> {ancestor's table rows, its log-likelihood}; everything after that runs out of LDS.
__device__ __forceinline__ void pk_book_particle(const pk_rank_args& a, int kg, bool local, const pk_book_lds& L) {
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < 64;
    const int n = a.n, N = a.N, k = kg - a.k0;
    const int grp = a.group_seeds ? kg / a.Kg : 0;       // batched independent sweeps: my group, its seed, my index in it
    const int gbase = grp * a.Kg;
    const uint64_t seed = a.group_seeds ? a.group_seeds[grp] : a.seed;
    const uint32_t kin = (uint32_t)(kg - gbase);
    if (w0) {
        // independent of the resampling outcome: issue first (slots >= 64 are copied further down)
        const double hb_l0 = (local && lane <= a.r) ? a.bl[(size_t)lane * a.Kloc + k] : 0.0;
        const double hb_r0 = (local && lane <= a.r) ? a.br[(size_t)lane * a.Kloc + k] : 0.0;
        const double ldf0 = (lane <= a.ldf_n) ? a.ldf[lane] : 0.0;
        // the pair keys (one Philox block per four slots) and the resampling draw: ONE evaluation for the wave when the
        // key blocks leave lane 63 free -- lanes 0..nb-1 take the key blocks, lane 63 the resampling counter
        const int nb = (n + 3) / 4;
        uint64_t Rdraw = 0;
        if (nb <= 63) {
            const bool res = lane == 63;
            const pm_u32x4 x = pm_philox4x32(kin, (uint32_t)a.r, res ? PM_STREAM_RESAMPLE : PM_STREAM_PAIR, res ? 0u : (uint32_t)lane, seed);
            if (lane < nb) { L.key[lane * 4 + 0] = x.x; L.key[lane * 4 + 1] = x.y; L.key[lane * 4 + 2] = x.z; L.key[lane * 4 + 3] = x.w; }
            Rdraw = ((uint64_t)(uint32_t)__shfl((int)x.y, 63, 64) << 32) | (uint32_t)__shfl((int)x.x, 63, 64);
        } else {
#pragma unroll 1
            for (int b = lane; b < nb; b += 64) {
                const pm_u32x4 x = pm_philox4x32(kin, (uint32_t)a.r, PM_STREAM_PAIR, (uint32_t)b, seed);
                L.key[b * 4 + 0] = x.x; L.key[b * 4 + 1] = x.y; L.key[b * 4 + 2] = x.z; L.key[b * 4 + 3] = x.w;
            }
            const pm_u32x4 x = pm_philox4x32(kin, (uint32_t)a.r, PM_STREAM_RESAMPLE, 0u, seed);
            Rdraw = ((uint64_t)x.y << 32) | x.x;
        }
        int anc = kg;
        if (a.r > 0) {
            const uint64_t R = Rdraw;
            if (a.flag) {                             // the scan runs in workgroup 0 of this launch: wait for its flag
                unsigned int spins = 0;
                while (__hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) {       // ~ seconds: give up loudly instead of hanging the GPU
                        if (lane == 0) __hip_atomic_store(a.timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            anc = gbase + pk_cdf_search_wave(a.cdf + gbase, a.group_seeds ? a.Kg : a.K, R, lane, a.flag != nullptr);
        }
        const int32_t* ro = a.roots_old + (size_t)anc * N;
        const int32_t* co = a.cnt_old + (size_t)anc * N;
        const double* rl = a.rootll_old + (size_t)anc * N;
        if (a.tab_ptrs) {                              // the owner of the ancestor holds its rows
            const char* base = a.tab_ptrs[anc / a.Kloc];
            ro = reinterpret_cast<const int32_t*>(base + a.tab_off_roots) + (size_t)anc * N;
            co = reinterpret_cast<const int32_t*>(base + a.tab_off_cnt) + (size_t)anc * N;
            rl = reinterpret_cast<const double*>(base + a.tab_off_rootll) + (size_t)anc * N;
        }
        #pragma unroll 1
        for (int i = lane; i < n; i += 64) { L.ro[i] = ro[i]; L.co[i] = co[i]; L.anc_ll[i] = rl[i]; }
        if (lane == 0) {
            L.misc[3] = anc;
            if (local) L.aux[AUX_LL_TILDE] = (a.r > 0) ? a.ll_prev[anc] : a.ll_tilde0;
        }
                if (lane <= a.r) { L.hbl[lane] = hb_l0; L.hbr[lane] = hb_r0; }
        if (lane <= a.ldf_n) L.ldf[lane] = ldf0;
        #pragma unroll 1
        for (int j = lane + 64; j <= a.r; j += 64) {
            L.hbl[j] = local ? a.bl[(size_t)j * a.Kloc + k] : 0.0;
            L.hbr[j] = local ? a.br[(size_t)j * a.Kloc + k] : 0.0;
        }
        #pragma unroll 1
        for (int j = lane + 64; j <= a.ldf_n; j += 64) L.ldf[j] = a.ldf[j];
    }
    __syncthreads();
    if (w0) {
        // largest key (lower slot on ties), then the second largest
        unsigned long long best = 0ull;
        #pragma unroll 1
        for (int i = lane; i < n; i += 64) {
            const unsigned long long c = ((unsigned long long)L.key[i] << 32) | (0xffffffffu - (uint32_t)i);
            best = c > best ? c : best;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const int il = (int)(0xffffffffu - (uint32_t)best);
        best = 0ull;
        #pragma unroll 1
        for (int i = lane; i < n; i += 64) {
            const unsigned long long c = ((unsigned long long)L.key[i] << 32) | (0xffffffffu - (uint32_t)i);
            if (i != il) best = c > best ? c : best;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const int ir = (int)(0xffffffffu - (uint32_t)best);
        // position of every remaining slot in ascending (key, slot) order
        int32_t* rn = a.roots_new + (size_t)kg * N;
        int32_t* cn = a.cnt_new + (size_t)kg * N;
        double* rln = a.rootll_new + (size_t)kg * N;
        #pragma unroll 1
        for (int i = lane; i < n; i += 64) {
            if (i == il || i == ir) continue;
            const unsigned long long mine = ((unsigned long long)L.key[i] << 32) | (uint32_t)i;
            int rank = 0;
            #pragma unroll 1
            for (int j = 0; j < n; ++j) {
                const unsigned long long cj = ((unsigned long long)L.key[j] << 32) | (uint32_t)j;
                rank += (j != il && j != ir && cj < mine) ? 1 : 0;
            }
            const int node = L.ro[i], c = L.co[i];
            const double xll = L.anc_ll[i];
            rn[rank] = node;
            cn[rank] = c;
            rln[rank] = xll;
            if (a.pos_hist) a.pos_hist[(size_t)kg * N + i] = rank;
            L.ord_cnt[rank] = c;
            L.ord_ll[rank] = xll;
            L.ord_ldf[rank] = L.ldf[c < a.ldf_n ? c : a.ldf_n];
        }
        if (lane == 0) {
            const int cnew = L.co[il] + L.co[ir];
            rn[n - 2] = N + a.r * a.K + kg;           // id of the node this particle creates now
            cn[n - 2] = cnew;
            L.ord_cnt[n - 2] = cnew;
            L.ord_ldf[n - 2] = L.ldf[cnew < a.ldf_n ? cnew : a.ldf_n];
            L.misc[0] = L.ro[il];
            L.misc[1] = L.ro[ir];
            if (a.pos_hist) { a.pos_hist[(size_t)kg * N + il] = -1; a.pos_hist[(size_t)kg * N + ir] = -1; }
            if (local) {
                a.merges[((size_t)a.r * a.Kloc + k) * 2 + 0] = il;
                a.merges[((size_t)a.r * a.Kloc + k) * 2 + 1] = ir;
                if (a.r > 0) a.ancestors[(size_t)(a.r - 1) * a.Kloc + k] = L.misc[3] - gbase;   // index inside the group
            }
        }
    }
    __syncthreads();
    if (tid == 0 && local) {                          // sequential sums, LDS operands only
        double sum_rem = 0.0, fprior = 0.0;
        int vminus = 0;
        #pragma unroll 1
        for (int p = 0; p < n - 2; ++p) sum_rem = sum_rem + L.ord_ll[p];
        #pragma unroll 1
        for (int p = 0; p < n - 1; ++p) {
            const int c = L.ord_cnt[p];
            fprior = fprior + (-L.ord_ldf[p]);
            vminus += c - (c == 1 ? 1 : 0);
        }
        double lp = 0.0, rp = 0.0;                    // history rows 0..r with THIS rank's rate (quirk Q3)
        #pragma unroll 1
        for (int j = 0; j <= a.r; ++j) {
            lp = lp + ((-a.lam_l) * L.hbl[j] + a.loglam_l);
            rp = rp + ((-a.lam_r) * L.hbr[j] + a.loglam_r);
        }
        const double b_l = L.hbl[a.r], b_r = L.hbr[a.r];
        const double q = 1.0 / ((double)((n - 1) * n) / 2.0);      // 1 / ncr(n, 2), vcsmc.py:298
        L.aux[AUX_SUM_REM] = sum_rem;
        L.aux[AUX_FPRIOR] = fprior;
        L.aux[AUX_LPRIOR] = lp;
        L.aux[AUX_RPRIOR] = rp;
        L.aux[AUX_PAREN] = ((a.loglam_l - a.lam_l * b_l) + a.loglam_r) - a.lam_r * b_r;
        L.aux[AUX_LOGV] = pm_log((double)vminus);
        L.aux[AUX_Q] = (a.flags & 1u) ? q : pm_log(q);
    }
    __syncthreads();
}

// ---- packed bookkeeping: LP lanes per particle, 64 / LP particles per wave ---------------------------------------
// The bookkeeping of one particle is a few short loops over its n <= N root slots; with one wave per particle most
// lanes idle and the kernel is bound by instruction issue (profiles/r01_merge_pmc.md: 656 VALU + 620 SALU per
// particle).  For N <= 32 a particle gets LP = 16 or 32 lanes, so one instruction stream serves 4 or 2 particles.
// Same arithmetic, same orders (the sequential sums stay sequential, on the first lane of each group).
template <int LP>
__device__ __forceinline__ int pk_cdf_search_group(const uint64_t* cdf, int K, uint64_t R, int sl, int lane) {
    const int gshift = lane & ~(LP - 1);
    const unsigned long long gmask = LP == 64 ? ~0ull : ((1ull << (LP & 63)) - 1ull);
    int lo = 0, hi = K;             // invariant: answer in [lo, hi), cdf[hi-1] > thr
    int step = (K + LP - 1) / LP;
    int p = (sl + 1) * step - 1;
    if (p > K - 1) p = K - 1;
    const uint64_t total = cdf[K - 1];
    uint64_t c = cdf[p];
    const uint64_t thr = pm_mulhi64(R, total);
    bool done = false;
    for (;;) {
        const unsigned long long ball = __ballot(!done && c > thr);
        if (!done) {
            const unsigned long long mask = (ball >> gshift) & gmask;
            const int f = mask ? __ffsll((long long)mask) - 1 : LP - 1;
            int pf = lo + (f + 1) * step - 1;
            if (pf > hi - 1) pf = hi - 1;
            lo = lo + f * step;
            hi = pf + 1;
            if (lo >= hi) lo = hi - 1;
            if (hi - lo <= 1) done = true;
            else {
                step = (hi - lo + LP - 1) / LP;
                p = lo + (sl + 1) * step - 1;
                if (p > hi - 1) p = hi - 1;
                c = cdf[p];
            }
        }
        if (__all(done)) break;
    }
    return lo;
}

template <int LP>
__device__ __forceinline__ void pk_book_packed(const pk_rank_args& a, int kg, bool local, const pk_book_lds& L, int sl, int lane) {
    const int n = a.n, N = a.N, k = kg - a.k0;
    const int grp = a.group_seeds ? kg / a.Kg : 0;
    const int gbase = grp * a.Kg;
    const uint64_t seed = a.group_seeds ? a.group_seeds[grp] : a.seed;
    const uint32_t kin = (uint32_t)(kg - gbase);
    {
        #pragma unroll 1
        for (int j = sl; j <= a.r; j += LP) {
            L.hbl[j] = local ? a.bl[(size_t)j * a.Kloc + k] : 0.0;
            L.hbr[j] = local ? a.br[(size_t)j * a.Kloc + k] : 0.0;
        }
        #pragma unroll 1
        for (int j = sl; j <= a.ldf_n; j += LP) L.ldf[j] = a.ldf[j];
        // pair keys (lanes 0..nb-1 of the group, nb <= LP / 4) and the resampling draw (last lane): one Philox evaluation
        const int nb = (n + 3) / 4;
        const bool res = sl == LP - 1;
        const pm_u32x4 x = pm_philox4x32(kin, (uint32_t)a.r, res ? PM_STREAM_RESAMPLE : PM_STREAM_PAIR, res ? 0u : (uint32_t)sl, seed);
        if (sl < nb) { L.key[sl * 4 + 0] = x.x; L.key[sl * 4 + 1] = x.y; L.key[sl * 4 + 2] = x.z; L.key[sl * 4 + 3] = x.w; }
        const int src = (lane & ~(LP - 1)) + LP - 1;
        const uint64_t Rdraw = ((uint64_t)(uint32_t)__shfl((int)x.y, src, 64) << 32) | (uint32_t)__shfl((int)x.x, src, 64);
        int anc = kg;
        if (a.r > 0) anc = gbase + pk_cdf_search_group<LP>(a.cdf + gbase, a.group_seeds ? a.Kg : a.K, Rdraw, sl, lane);
        const int32_t* ro = a.roots_old + (size_t)anc * N;
        const int32_t* co = a.cnt_old + (size_t)anc * N;
        const double* rl = a.rootll_old + (size_t)anc * N;
        if (a.tab_ptrs) {                              // the owner of the ancestor holds its rows
            const char* base = a.tab_ptrs[anc / a.Kloc];
            ro = reinterpret_cast<const int32_t*>(base + a.tab_off_roots) + (size_t)anc * N;
            co = reinterpret_cast<const int32_t*>(base + a.tab_off_cnt) + (size_t)anc * N;
            rl = reinterpret_cast<const double*>(base + a.tab_off_rootll) + (size_t)anc * N;
        }
        #pragma unroll 1
        for (int i = sl; i < n; i += LP) { L.ro[i] = ro[i]; L.co[i] = co[i]; L.anc_ll[i] = rl[i]; }
        if (sl == 0) {
            L.misc[3] = anc;
            if (local) L.aux[AUX_LL_TILDE] = (a.r > 0) ? a.ll_prev[anc] : a.ll_tilde0;
        }
    }
    __syncthreads();
    {
        // largest key (lower slot on ties), then the second largest
        unsigned long long best = 0ull;
        #pragma unroll 1
        for (int i = sl; i < n; i += LP) {
            const unsigned long long c = ((unsigned long long)L.key[i] << 32) | (0xffffffffu - (uint32_t)i);
            best = c > best ? c : best;
        }
#pragma unroll
        for (int off = 1; off < LP; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const int il = (int)(0xffffffffu - (uint32_t)best);
        best = 0ull;
        #pragma unroll 1
        for (int i = sl; i < n; i += LP) {
            const unsigned long long c = ((unsigned long long)L.key[i] << 32) | (0xffffffffu - (uint32_t)i);
            if (i != il) best = c > best ? c : best;
        }
#pragma unroll
        for (int off = 1; off < LP; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const int ir = (int)(0xffffffffu - (uint32_t)best);
        // position of every remaining slot in ascending (key, slot) order
        int32_t* rn = a.roots_new + (size_t)kg * N;
        int32_t* cn = a.cnt_new + (size_t)kg * N;
        double* rln = a.rootll_new + (size_t)kg * N;
        #pragma unroll 1
        for (int i = sl; i < n; i += LP) {
            if (i == il || i == ir) continue;
            const unsigned long long mine = ((unsigned long long)L.key[i] << 32) | (uint32_t)i;
            int rank = 0;
            #pragma unroll 1
            for (int j = 0; j < n; ++j) {
                const unsigned long long cj = ((unsigned long long)L.key[j] << 32) | (uint32_t)j;
                rank += (j != il && j != ir && cj < mine) ? 1 : 0;
            }
            const int node = L.ro[i], c = L.co[i];
            const double xll = L.anc_ll[i];
            rn[rank] = node;
            cn[rank] = c;
            rln[rank] = xll;
            if (a.pos_hist) a.pos_hist[(size_t)kg * N + i] = rank;
            L.ord_cnt[rank] = c;
            L.ord_ll[rank] = xll;
            L.ord_ldf[rank] = L.ldf[c < a.ldf_n ? c : a.ldf_n];
        }
        if (sl == 0) {
            const int cnew = L.co[il] + L.co[ir];
            rn[n - 2] = N + a.r * a.K + kg;           // id of the node this particle creates now
            cn[n - 2] = cnew;
            L.ord_cnt[n - 2] = cnew;
            L.ord_ldf[n - 2] = L.ldf[cnew < a.ldf_n ? cnew : a.ldf_n];
            L.misc[0] = L.ro[il];
            L.misc[1] = L.ro[ir];
            if (a.pos_hist) { a.pos_hist[(size_t)kg * N + il] = -1; a.pos_hist[(size_t)kg * N + ir] = -1; }
            if (local) {
                a.merges[((size_t)a.r * a.Kloc + k) * 2 + 0] = il;
                a.merges[((size_t)a.r * a.Kloc + k) * 2 + 1] = ir;
                if (a.r > 0) a.ancestors[(size_t)(a.r - 1) * a.Kloc + k] = L.misc[3] - gbase;   // index inside the group
            }
        }
    }
    __syncthreads();
    if (local) {
        // The sequential sums of the contract, LDS operands only.  Four independent chains -- the remaining roots' log-likelihoods,
        // their leaf-count priors, the left and the right branch-history priors with THIS rank's rate (quirk Q3) -- run on four
        // lanes of the particle's group through ONE loop, acc += a x[j] + b with the chain's own (x, a, b, length): a = 1, b = 0
        // and a = -1, b = 0 reproduce `acc + x` and `acc + (-x)` bit for bit.  (One lane used to walk all four in turn: half of
        // this kernel's instructions, for 4 of 64 lanes.)
        const double* src = sl == 0 ? L.ord_ll : sl == 1 ? L.ord_ldf : sl == 2 ? L.hbl : L.hbr;
        const int cnt = sl == 0 ? n - 2 : sl == 1 ? n - 1 : sl < 4 ? a.r + 1 : 0;
        const double ca = sl == 0 ? 1.0 : sl == 1 ? -1.0 : sl == 2 ? -a.lam_l : -a.lam_r;
        const double cb = sl == 2 ? a.loglam_l : sl == 3 ? a.loglam_r : 0.0;
        const int trips = n - 1 > a.r + 1 ? n - 1 : a.r + 1;
        double acc = 0.0;
        #pragma unroll 1
        for (int j = 0; j < trips; ++j)
            if (j < cnt) acc = acc + (ca * src[j] + cb);
        int vminus = 0;                                   // integers: any order
        #pragma unroll 1
        for (int p = sl; p < n - 1; p += LP) {
            const int c = L.ord_cnt[p];
            vminus += c - (c == 1 ? 1 : 0);
        }
#pragma unroll
        for (int off = 1; off < LP; off <<= 1) vminus += __shfl_xor(vminus, off, 64);
        const double q = 1.0 / ((double)((n - 1) * n) / 2.0);      // 1 / ncr(n, 2), vcsmc.py:298
        const double lg = pm_log(sl == 0 ? (double)vminus : q);    // lane 0: log v-, lane 1: log q (one instruction stream)
        if (sl == 0) {
            const double b_l = L.hbl[a.r], b_r = L.hbr[a.r];
            L.aux[AUX_SUM_REM] = acc;
            L.aux[AUX_PAREN] = ((a.loglam_l - a.lam_l * b_l) + a.loglam_r) - a.lam_r * b_r;
            L.aux[AUX_LOGV] = lg;
        } else if (sl == 1) {
            L.aux[AUX_FPRIOR] = acc;
            L.aux[AUX_Q] = (a.flags & 1u) ? q : lg;
        } else if (sl == 2) {
            L.aux[AUX_LPRIOR] = acc;
        } else if (sl == 3) {
            L.aux[AUX_RPRIOR] = acc;
        }
    }
    __syncthreads();
}

// ---- Sharded sweep: the local cache of remote nodes.  Over xGMI a remote row is fetched again by every launch that touches it (the
//      mapping of a peer's memory is cached in L2 for the length of a kernel at best) and by every XCD, while the genealogy makes
//      thousands of particles merge the same few nodes: the FIRST particle of this rank whose bookkeeping picks a remote node as a
//      child claims its mirror entry (compare-and-swap), takes a slot and its wave copies the row, once per sweep; every later merge
//      (this rank event's included: the copy is complete when the bookkeeping launch ends) reads the local copy (pk_node_ptr).
//      The cache is bounded: when it is full the entry says so (-2) and the node is read in place as before.  Same bits either way.
__device__ __forceinline__ bool pk_remote_node(const pk_rank_args& a, int id, int& x, const double*& src) {
    if (id < a.N) return false;
    x = id - a.N;
    const int rho = x / a.K, kap = x - rho * a.K, owner = kap / a.Kloc;
    if (owner * a.Kloc == a.k0) return false;
    src = a.pool_ptrs[owner] + ((size_t)rho * a.Kloc + (kap - owner * a.Kloc)) * (size_t)a.S * 4;
    return true;
}
// one lane: claim node `id` for this rank's cache; the slot to fill, or -1 (local, a leaf, somebody else's, or no room)
__device__ __forceinline__ int pk_cache_claim(const pk_rank_args& a, int id) {
    int x;
    const double* src;
    if (!pk_remote_node(a, id, x, src)) return -1;
    if (__hip_atomic_load(a.mirror + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;
    int expected = 0;
    if (!__hip_atomic_compare_exchange_strong(a.mirror + x, &expected, -1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return -1;
    const int slot = __hip_atomic_fetch_add(a.mirror + (size_t)(a.N - 1) * a.K, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot >= a.cache_cap) {
        __hip_atomic_store(a.mirror + x, -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
    }
    return slot;
}
// the whole wave: copy node `id` (wave-uniform, claimed with slot `slot`) into the cache, then publish the slot
__device__ __forceinline__ void pk_cache_fill(const pk_rank_args& a, int id, int slot, int lane) {
    typedef __attribute__((ext_vector_type(4))) unsigned int cu4;
    typedef __attribute__((address_space(1))) const cu4 gsrc_t;
    typedef __attribute__((address_space(1))) cu4 gdst_t;
    int x;
    const double* srcd;
    pk_remote_node(a, id, x, srcd);
    const char* src = (const char*)srcd;
    char* dst = (char*)(a.cache + (size_t)slot * (size_t)a.S * 4);
    const size_t bytes = (size_t)a.S * 32;
    size_t o = (size_t)lane * 16;
    for (; o + 3 * 1024 < bytes; o += 4 * 1024) {          // four 16-byte loads per lane in flight (a remote round trip each)
        const cu4 v0 = *(gsrc_t*)(src + o), v1 = *(gsrc_t*)(src + o + 1024), v2 = *(gsrc_t*)(src + o + 2048), v3 = *(gsrc_t*)(src + o + 3072);
        *(gdst_t*)(dst + o) = v0; *(gdst_t*)(dst + o + 1024) = v1; *(gdst_t*)(dst + o + 2048) = v2; *(gdst_t*)(dst + o + 3072) = v3;
    }
    for (; o < bytes; o += 1024) *(gdst_t*)(dst + o) = *(gsrc_t*)(src + o);
    if (lane == 0) __hip_atomic_store(a.mirror + x, slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int LP>
__global__ __launch_bounds__(64) void pk_rank_book_packed(const pk_rank_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, sub = lane / LP, sl = lane & (LP - 1);
    const int count = a.tab_ptrs ? a.Kloc : a.K;                      // local bookkeeping: this rank's particles only
    int idx = blockIdx.x * (64 / LP) + sub;
    if (idx >= count) idx = count - 1;          // a spare group repeats the last particle: identical values, identical writes
    const int kg = a.tab_ptrs ? a.k0 + idx : idx;
    const pk_book_lds L = pk_book_carve(smem + (size_t)sub * pk_book_lds_bytes(a.N), a.N);
    const bool local = kg >= a.k0 && kg < a.k0 + a.Kloc;
    pk_book_packed<LP>(a, kg, local, L, sl, lane);
    if (local) {
        const int k = kg - a.k0;
#pragma unroll
        for (int t = sl; t < PK_AUX + 2; t += LP) {     // (one trip for LP >= 16)
            if (t < PK_AUX) a.aux[(size_t)k * PK_AUX + t] = L.aux[t];
            else a.child[k * 2 + (t - PK_AUX)] = L.misc[t - PK_AUX];
        }
    }
    if (a.lazy && a.r > 0 && sl == 0) {
        const int anc = L.misc[3];
        a.mark[(size_t)(a.r - 1) * a.K + anc] = 1u;   // plain store: every adopter writes the same value
    }
    if (a.mirror && a.r > 0) {                        // (uniform) sharded: new remote children into the local cache, by the whole wave
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = L.misc[c];
            const int slot = (local && sl == 0) ? pk_cache_claim(a, id) : -1;
            for (int g2 = 0; g2 < 64 / LP; ++g2) {
                const int s2 = __shfl(slot, g2 * LP, 64), id2 = __shfl(id, g2 * LP, 64);
                if (s2 >= 0) pk_cache_fill(a, id2, s2, lane);
            }
        }
    }
}

// Sharded lazy nodes: every rank derives the resampling outcome of ALL K particles (index search only, no tables)
// and marks the adopted ancestors' nodes of the previous rank event, so that each owner knows which of its nodes
// to write before anybody merges them.  16 lanes per particle.
__global__ __launch_bounds__(64) void pk_all_marks(const pk_rank_args a) {
    const int lane = threadIdx.x, sub = lane >> 4, sl = lane & 15;
    int kg = blockIdx.x * 4 + sub;
    if (kg >= a.K) kg = a.K - 1;
    const int grp = a.group_seeds ? kg / a.Kg : 0;
    const int gbase = grp * a.Kg;
    const uint64_t seed = a.group_seeds ? a.group_seeds[grp] : a.seed;
    const pm_u32x4 x = pm_philox4x32((uint32_t)(kg - gbase), (uint32_t)a.r, PM_STREAM_RESAMPLE, 0u, seed);
    const uint64_t R = ((uint64_t)x.y << 32) | x.x;
    const int anc = gbase + pk_cdf_search_group<16>(a.cdf + gbase, a.group_seeds ? a.Kg : a.K, R, sl, lane);
    if (sl == 0) a.mark[(size_t)(a.r - 1) * a.K + anc] = 1u;
}

// Write node (rho, kappa) into the pool: the same merge, row per thread, no likelihood.  Called by ONE wave.
__device__ __forceinline__ const double* pk_node_ptr(const pk_rank_args& a, int id);
template <int U = 4>
__device__ __forceinline__ void pk_materialize_node(const pk_rank_args& a, int rho, int kappa, int lane, int nthreads,
                                                    int s_begin, int s_end) {
    const int k = kappa - a.k0;
    const int32_t* ch = a.child_all + ((size_t)rho * a.Kloc + k) * 2;
    const double* Lp = pk_node_ptr(a, ch[0]);
    const double* Rp = pk_node_ptr(a, ch[1]);
    const double* P = a.Pmat_all + ((size_t)rho * a.Kloc + k) * 32;
    double Pl[16], Pr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { Pl[j] = P[j]; Pr[j] = P[16 + j]; }
    double* out = a.pool + ((size_t)rho * a.Kloc + k) * (size_t)a.S * 4;
    // U rows per thread in flight (four in the kernels that only write nodes): the rows are independent, and a node of a few
    // hundred rows is otherwise one dependent load -> store round trip after the other (a short alignment is then read in ONE
    // round trip).  The kernels that also do the bookkeeping have no registers to spare for it (U = 1).
    for (int s = s_begin + lane; s < s_end; s += U * nthreads) {
        double Lv[U][4], Rv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int su = s + u * nthreads < s_end ? s + u * nthreads : s;      // (a row past the end re-reads this thread's first)
            pk_load4(Lp + (size_t)su * 4, Lv[u]);
            pk_load4(Rp + (size_t)su * 4, Rv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double o[4];
            pk_merge_site(Lv[u], Rv[u], Pl, Pr, o);
            if (s + u * nthreads < s_end) pk_store4(out + (size_t)(s + u * nthreads) * 4, o);
        }
    }
}

// After the bookkeeping of rank event r: the nodes of rank event r-1 whose creator was adopted by some particle
// are now live and get written; their children are leaves or nodes that were adopted, hence written, at an
// earlier rank event.  Grid (site tiles, K): rows beyond the number of queued nodes exit at once (few distinct
// ancestors survive a resampling); a node is spread over ceil(S / 1024) workgroups so that a large node is
// not limited to one CU's bandwidth.
#define PK_MAT_TILE 1024
__global__ __launch_bounds__(PK_COLS) void pk_materialize_adopted(const pk_rank_args a) {
    const int node = a.k0 + blockIdx.y;                // grid (site tiles, local particles): unmarked nodes leave at once
    if (!a.mark[(size_t)(a.r - 1) * a.K + node]) return;
    const int tile = (a.S + gridDim.x - 1) / gridDim.x;  // the host picks the tile count (one tile for small nodes)
    const int s0 = blockIdx.x * tile, s1 = s0 + tile < a.S ? s0 + tile : a.S;
    pk_materialize_node(a, a.r - 1, node, threadIdx.x, PK_COLS, s0, s1);
}

// The same for small nodes, where nearly every workgroup of the grid above would start only to read one mark and leave
// (dispatching 20 480 such workgroups costs 8 us): a workgroup takes PK_MAT_GROUP particles, one wave reads their marks with one
// coalesced load, and the marked nodes -- a handful on real data -- are written one after the other by the whole workgroup.
#define PK_MAT_GROUP 64
__global__ __launch_bounds__(PK_COLS) void pk_materialize_adopted_grouped(const pk_rank_args a) {
    __shared__ unsigned long long marked;
    const int k0 = blockIdx.x * PK_MAT_GROUP;                   // local particle slots k0 .. k0 + PK_MAT_GROUP - 1
    if (threadIdx.x < 64) {
        const int k = k0 + (int)threadIdx.x;
        const unsigned int m = (threadIdx.x < PK_MAT_GROUP && k < a.Kloc) ? a.mark[(size_t)(a.r - 1) * a.K + a.k0 + k] : 0u;
        const unsigned long long b = __ballot(m != 0u);
        if (threadIdx.x == 0) marked = b;
    }
    __syncthreads();
    unsigned long long todo = marked;
    while (todo) {                                              // workgroup-uniform
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        pk_materialize_node(a, a.r - 1, a.k0 + k0 + j, threadIdx.x, PK_COLS, 0, a.S);
    }
}

// every node of rank event rho (test surface: phylo_sweep_node after a lazy sweep)
__global__ __launch_bounds__(PK_COLS) void pk_materialize_rank(const pk_rank_args a, int rho) {
    const int k = blockIdx.x;
    if (a.mark[(size_t)rho * a.K + a.k0 + k]) return;
    pk_materialize_node(a, rho, a.k0 + k, threadIdx.x, PK_COLS, 0, a.S);
    __syncthreads();
    if (threadIdx.x == 0) a.mark[(size_t)rho * a.K + a.k0 + k] = 1u;
}

// every node of rank event rho, unconditionally (the nodes of the last rank event are not stored by the sweep)
__global__ __launch_bounds__(PK_COLS) void pk_materialize_all(const pk_rank_args a, int rho) {
    pk_materialize_node(a, rho, a.k0 + blockIdx.x, threadIdx.x, PK_COLS, 0, a.S);
}

// Bookkeeping AND the writes of the adopted nodes in one launch (one GPU, lazy nodes, N <= 64: 16, 32 or 64 lanes per particle).  Launched one after the other
// they are two dependent launches at their latency floors (7.4 + 5.1 us at K = 2048): pk_materialize_adopted waits for the marks
// the bookkeeping leaves.  But whether particle k was adopted does not need the bookkeeping: with the K draws of this rank event
// (written once per sweep by pk_sweep_prologue), thr = mulhi64(draw, cdf total) is what the index search compares the cdf with, and k
// is adopted iff some thr lies in [cdf[k-1], cdf[k]).
// So workgroups [0, book_blocks) do the packed bookkeeping (256 / LP particles each) while workgroups behind them test the
// thresholds and write node (r-1, k) where needed:
//   per_particle: one workgroup per particle (K / 256 thresholds per thread, most workgroups leave after the test);
//   grouped (large batched launches): one workgroup per 64 particles, hits located in the group's 64 cdf values, written in turn.
// Nodes written here are read by the merge launch that follows, never inside this launch.  Same values, same bits.
__device__ __forceinline__ void pk_mat_by_thresholds(const pk_rank_args& a, int kg) {
    const int tid = threadIdx.x;
    const int grp = a.group_seeds ? kg / a.Kg : 0, gbase = grp * a.Kg, Kg = a.group_seeds ? a.Kg : a.K, kin = kg - gbase;
    const unsigned long long lo = kin > 0 ? a.cdf[gbase + kin - 1] : 0ull, hi = a.cdf[gbase + kin], total = a.cdf[gbase + Kg - 1];
    unsigned int hit = 0u;                                // no branch around the loads: they all travel together
    #pragma unroll 4
    for (int i = tid; i < Kg; i += PK_COLS) {
        const unsigned long long t = pm_mulhi64(a.rdraw[gbase + i], total);
        hit |= (unsigned int)(t >= lo) & (unsigned int)(t < hi);
    }
    if (!__syncthreads_or((int)hit)) return;
    pk_materialize_node<1>(a, a.r - 1, kg, tid, PK_COLS, 0, a.S);
    if (tid == 0) a.mark[(size_t)(a.r - 1) * a.K + kg] = 1u;
}
__device__ __forceinline__ void pk_mat_by_thresholds_grouped(const pk_rank_args& a, int k0) {   // particles k0 .. k0 + 63, one group
    __shared__ unsigned long long gcdf[PK_MAT_GROUP];
    __shared__ unsigned int maskw[2];
    const int tid = threadIdx.x;
    const int grp = a.group_seeds ? k0 / a.Kg : 0, gbase = grp * a.Kg, Kg = a.group_seeds ? a.Kg : a.K, kin0 = k0 - gbase;
    const int cnt = Kg - kin0 < PK_MAT_GROUP ? Kg - kin0 : PK_MAT_GROUP;
    if (tid < PK_MAT_GROUP) gcdf[tid] = a.cdf[gbase + kin0 + (tid < cnt ? tid : cnt - 1)];
    if (tid < 2) maskw[tid] = 0u;
    const unsigned long long lo = kin0 > 0 ? a.cdf[gbase + kin0 - 1] : 0ull, total = a.cdf[gbase + Kg - 1];
    __syncthreads();
    const unsigned long long hi = gcdf[cnt - 1];
    // a heavy particle attracts most of the Kg draws, so hits are NOT rare: every thread collects its hits in a register mask
    // (binary search of the 64 cdf values in LDS), the masks are OR-ed across the wave by a butterfly and across waves in LDS
    unsigned long long mine = 0ull;
    #pragma unroll 2
    for (int i = tid; i < Kg; i += PK_COLS) {
        const unsigned long long t = pm_mulhi64(a.rdraw[gbase + i], total);
        if (t >= lo && t < hi) {                          // first j with gcdf[j] > t
            int l = 0, h = cnt - 1;
            while (l < h) {
                const int m = (l + h) >> 1;
                if (gcdf[m] > t) h = m; else l = m + 1;
            }
            mine |= 1ull << l;
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mine |= __shfl_xor(mine, off, 64);
    if ((tid & 63) == 0 && mine) {
        if ((unsigned int)mine) atomicOr(&maskw[0], (unsigned int)mine);            // four waves, integer OR: any order, same result
        if ((unsigned int)(mine >> 32)) atomicOr(&maskw[1], (unsigned int)(mine >> 32));
    }
    __syncthreads();
    unsigned long long todo = ((unsigned long long)maskw[1] << 32) | maskw[0];
    while (todo) {                                        // workgroup-uniform
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        pk_materialize_node<1>(a, a.r - 1, k0 + j, tid, PK_COLS, 0, a.S);
        if (tid == 0) a.mark[(size_t)(a.r - 1) * a.K + k0 + j] = 1u;
    }
}
// the same as a launch of its own (sharded sweeps: each rank writes ITS adopted nodes before the barrier collective of the first
// half of a rank event); grid: local particles, or groups of 64 of them
__global__ __launch_bounds__(PK_COLS) void pk_materialize_by_draws(const pk_rank_args a, int grouped) {
    if (grouped) pk_mat_by_thresholds_grouped(a, a.k0 + (int)blockIdx.x * PK_MAT_GROUP);
    else pk_mat_by_thresholds(a, a.k0 + (int)blockIdx.x);
}

template <int LP>
__global__ __launch_bounds__(PK_COLS, 5) void pk_rank_book_mat(const pk_rank_args a, int book_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= book_blocks) {                 // (one workgroup per particle: the launch is a single sweep's)
        pk_mat_by_thresholds(a, (int)blockIdx.x - book_blocks);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, sub = tid / LP, sl = tid & (LP - 1);
    int idx = blockIdx.x * (PK_COLS / LP) + sub;
    if (idx >= a.K) idx = a.K - 1;              // a spare group repeats the last particle: identical values, identical writes
    const int kg = idx;                         // one GPU: every particle is local
    const pk_book_lds L = pk_book_carve(smem + (size_t)sub * pk_book_lds_bytes(a.N), a.N);
    pk_book_packed<LP>(a, kg, true, L, sl, lane);
    if (sl < PK_AUX + 2) {
        if (sl < PK_AUX) a.aux[(size_t)kg * PK_AUX + sl] = L.aux[sl];
        else a.child[kg * 2 + (sl - PK_AUX)] = L.misc[sl - PK_AUX];
    }
    if (a.r > 0 && sl == 0) a.mark[(size_t)(a.r - 1) * a.K + L.misc[3]] = 1u;   // the marks stay (phylo_sweep_node reads them)
}

// Bookkeeping kernel: one 64-thread workgroup (one wave) per GLOBAL particle.  Particles of this rank's
// shard also get their child node ids and weight terms written for the merge kernel; for the others only the
// replicated integer state (root tables) is advanced.
__global__ __launch_bounds__(64) void pk_rank_book(const pk_rank_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kg = a.tab_ptrs ? a.k0 + blockIdx.x : blockIdx.x;      // local bookkeeping: this rank's particles only
    const pk_book_lds L = pk_book_carve(smem, a.N);
    const bool local = kg >= a.k0 && kg < a.k0 + a.Kloc;
    pk_book_particle(a, kg, local, L);
    if (local && threadIdx.x < PK_AUX + 2) {
        const int k = kg - a.k0;
        if (threadIdx.x < PK_AUX) a.aux[(size_t)k * PK_AUX + threadIdx.x] = L.aux[threadIdx.x];
        else a.child[k * 2 + (threadIdx.x - PK_AUX)] = L.misc[threadIdx.x - PK_AUX];
    }
    if (a.lazy && a.r > 0 && threadIdx.x == 0) {
        // this particle adopted the table of `anc`: the node anc created at the previous rank event is now live;
        // pk_materialize_adopted writes the marked nodes of that rank event.
        const int anc = L.misc[3];                    // every rank sees every adoption; the OWNER of the node writes it
        a.mark[(size_t)(a.r - 1) * a.K + anc] = 1u;  // plain store: every adopter writes the same value (no contended atomics)
    }
    if (a.mirror && a.r > 0 && local) {               // (uniform) sharded: new remote children into the local cache
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = L.misc[c];
            const int slot = __shfl(threadIdx.x == 0 ? pk_cache_claim(a, id) : -1, 0, 64);
            if (slot >= 0) pk_cache_fill(a, id, slot, threadIdx.x);
        }
    }
}

// Scan + bookkeeping in ONE launch: workgroup 0 (256 threads) runs the resampling scan of log w_{r-1} and
// publishes the cdf; workgroups 1..K (their first wave) do the bookkeeping of particle blockIdx-1, overlapping
// everything that does not depend on the resampling outcome with the scan and polling the flag just before
// the index search.  Hand-off form: 8-byte agent-scope stores of the payload, every storing wave drained
// (s_waitcnt vmcnt(0)), workgroup barrier, ONE agent-scope flag store; consumers poll the flag relaxed and read
// the payload with agent-scope loads (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides").
// Workgroup 0 is dispatched first and waits for nobody, so the launch cannot deadlock; the poll is bounded anyway.
__global__ __launch_bounds__(PK_COLS) void pk_rank_scan_book(const pk_rank_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {
        pk_scan_lds* sh = reinterpret_cast<pk_scan_lds*>(smem);
        unsigned long long* stage = (a.K <= PK_SCAN_LDS_MAX_K) ? reinterpret_cast<unsigned long long*>(smem + sizeof(pk_scan_lds))
                                                               : reinterpret_cast<unsigned long long*>(a.scan_cdf);
        pk_scan_block(a.scan_logw, a.K, a.scan_cdf, a.scan_lse, sh, stage, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.flag, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x >= 64) return;                              // bookkeeping uses one wave
    const int kg = blockIdx.x - 1;
    const pk_book_lds L = pk_book_carve(smem, a.N);
    const bool local = kg >= a.k0 && kg < a.k0 + a.Kloc;
    pk_book_particle(a, kg, local, L);
    if (local && threadIdx.x < PK_AUX + 2) {
        const int k = kg - a.k0;
        if (threadIdx.x < PK_AUX) a.aux[(size_t)k * PK_AUX + threadIdx.x] = L.aux[threadIdx.x];
        else a.child[k * 2 + (threadIdx.x - PK_AUX)] = L.misc[threadIdx.x - PK_AUX];
    }
    if (a.lazy && a.r > 0 && threadIdx.x == 0) {
        // this particle adopted the table of `anc`: the node anc created at the previous rank event is now live;
        // pk_materialize_adopted writes the marked nodes of that rank event.
        const int anc = L.misc[3];                    // every rank sees every adoption; the OWNER of the node writes it
        a.mark[(size_t)(a.r - 1) * a.K + anc] = 1u;  // plain store: every adopter writes the same value (no contended atomics)
    }
}

__device__ __forceinline__ const double* pk_node_ptr(const pk_rank_args& a, int id) {
    const size_t node_sz = (size_t)a.S * 4;
    // a child is a leaf (id < N, replicated on every GPU) or the node id = N + rho*K + kappa created at rank
    // event rho by global particle kappa, which lives in the pool of rank kappa / Kloc (read in place over
    // xGMI when that is not this GPU)
    if (id < a.N) return a.leaves + (size_t)id * node_sz;
    if (a.Kloc == a.K) return a.pool + (size_t)(id - a.N) * node_sz;      // one rank: no dependent load of the owner's base
    const int x = id - a.N, rho = x / a.K, kap = x - rho * a.K, owner = kap / a.Kloc;
    if (a.mirror && owner * a.Kloc != a.k0) {              // a remote node this rank has fetched already: its local copy
        const int slot = a.mirror[x];
        if (slot > 0) return a.cache + (size_t)(slot - 1) * node_sz;
    }
    return a.pool_ptrs[owner] + ((size_t)rho * a.Kloc + (kap - owner * a.Kloc)) * node_sz;
}

// Leaf children whose rows are one-hot / all-ones (the reference's encoding, runner.py:83-96) are read as a
// 1-byte code per site; (row . P) is then a 16-byte LDS lookup -- bit-identical to the generic chain on
// those rows, without moving the 32-byte row.
// table[code][j] = (leaf row of that code . P)[j]: rows 0..3 are the rows of P, row 4 the chain over an all-ones row
__device__ __forceinline__ void pk_build_leaf_table(const double* __restrict__ P /*16, uniform*/, double (*tab)[4], int t) {
    if (t < 16) tab[t >> 2][t & 3] = P[t];
    else if (t < 20) {
        const int j = t - 16;
        tab[4][j] = pm_fma(1.0, P[12 + j], pm_fma(1.0, P[8 + j], pm_fma(1.0, P[4 + j], 1.0 * P[j])));
    }
}

// log_likelihood_r and log w_r of the particle from the new node's log-likelihood and the bookkeeping terms (k8)
__device__ __forceinline__ void pk_merge_epilogue(const pk_rank_args& a, int k, int kg, double tot) {
    const double* ax = a.aux + (size_t)k * PK_AUX;
    const double fl = ax[AUX_SUM_REM] + tot;
    const double ll = ((fl + ax[AUX_FPRIOR]) + ax[AUX_LPRIOR]) + ax[AUX_RPRIOR];
    const double lw = (((ll - ax[AUX_LL_TILDE]) - ax[AUX_PAREN]) + ax[AUX_LOGV]) - ax[AUX_Q];
    a.nodell[a.N + a.r * a.K + kg] = tot;
    a.rootll_new[(size_t)kg * a.N + (a.n - 2)] = tot;
    a.ll_r[kg] = ll;
    a.logw_r[kg] = lw;
}

// rows longer than one tile: the merge waves left their tiles' values in tilev[k][0 .. ntiles); added left to right
__global__ __launch_bounds__(256) void pk_tile_epilogue(const pk_rank_args a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.Kloc) return;
    const double* tv = a.tilev + (size_t)k * a.ntiles;
    double tot = tv[0];
    for (int t = 1; t < a.ntiles; ++t) tot = tot + tv[t];
    pk_merge_epilogue(a, k, a.k0 + k, tot);
}

// multi-GPU: after the all-gather of node log-likelihoods, complete the root tables of the other ranks'
// particles with the value of the node each of them created at this rank event
__global__ void pk_fix_rootll(double* __restrict__ rootll_new, const double* __restrict__ nodell_row, int K, int N, int n,
                              int k0, int Kloc) {
    const int kg = blockIdx.x * blockDim.x + threadIdx.x;
    if (kg >= K || (kg >= k0 && kg < k0 + Kloc)) return;
    rootll_new[(size_t)kg * N + (n - 2)] = nodell_row[kg];
}

// ================================================================================================
// T: the twisted / nested proposal (vncsmc.py:295-416, 432-499).  Per rank event:
//   pk_twist_adopt_draws  resampling index + adoption of the ancestor's root table (no pick yet), and -- independent of it, in the
//                       same launch -- branch lengths and transition matrices of every (particle, pair, sub-sample)
//   pk_twist_potentials look-ahead potential post(merged) - post(left) - post(right) of every one of them
//   pk_twist_choose     normalise per particle, draw ONE (pair, sub-sample), weight terms, children
//   pk_twist_tables     new root tables from the chosen pair (all particles; after the all-gather of the
//                       choices when sharded), remaining roots in DESCENDING slot order (vncsmc.py:305)
// then the ordinary pk_rank_merge.  Sub-sample j = t*M + m of pair t (lexicographic r1 < r2).
// ================================================================================================
#define PK_TWIST_MAX_M 1024
#define PK_TWIST_MAX_J (1 << 20)
#define PK_TWIST_LDS_J 8192             // sub-samples per particle whose weights live in LDS (pk_twist_choose, pg_twist_tau); more go through wbuf
#define PK_TWIST_DRAW_BLOCK 0xFFFFFFFFu

struct pk_twist_args {
    pk_rank_args a;                  // shared fields (tables, cdf, rates, outputs)
    int M, J;                        // J = C(n,2) * M
    int32_t* roots_ad; int32_t* cnt_ad; double* rootll_ad;   // [K][N] adopted (resampled) tables
    double* tw_b;                    // [Kloc][J][2] branch lengths
    double* tw_P;                    // [Kloc][J][32] transition matrices
    double* pot;                     // [Kloc][J] potentials
    double* chosen;                  // [K] chosen j per global particle (as double: travels with the RCCL all-gather)
    double* Pmat_r;                  // [Kloc][32] matrices of the chosen sub-sample (input of pk_rank_merge)
    double* bl_r; double* br_r;      // [Kloc] rows r of the branch-length history
    const uint32_t* pair_hist;       // [N][N][32] or NULL: sites per code pair (c_l * 5 + c_r) of two coded leaves
    const uint8_t* codes;            // [N][S] leaf codes or NULL: set when the DATA is coded (contracts v3, v4), whichever
                                     // access path the merge uses
    int own_tables;                  // pk_twist_choose also writes its particle's new root table (one GPU)
    double* wbuf;                    // [Kloc][J] softmax weights when J > PK_TWIST_LDS_J (else they live in LDS)
};

// sites per code pair of every ordered pair of coded leaves: grid (N, N).  Integer LDS atomics (exact).
__global__ __launch_bounds__(256) void pk_pair_hist(const uint8_t* __restrict__ codes, int N, int S, uint32_t* __restrict__ hist) {
    __shared__ unsigned int h[32];
    const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < 32) h[tid] = 0u;
    __syncthreads();
    const uint8_t* ca = codes + (size_t)a * S;
    const uint8_t* cb = codes + (size_t)b * S;
    for (int s = tid; s < S; s += 256) atomicAdd(&h[ca[s] * 5 + cb[s]], 1u);
    __syncthreads();
    if (tid < 32) hist[((size_t)a * N + b) * 32 + tid] = h[tid];
}

__device__ __forceinline__ void pk_twist_adopt_body(const pk_twist_args& ta, int kg) {
    const pk_rank_args& a = ta.a;
    const int lane = threadIdx.x, N = a.N, n = a.n;
    int anc = kg;
    if (a.r > 0) {
        const pm_u32x4 x = pm_philox4x32((uint32_t)kg, (uint32_t)a.r, PM_STREAM_RESAMPLE, 0u, a.seed);
        anc = pk_cdf_search_wave(a.cdf, a.K, ((uint64_t)x.y << 32) | x.x, lane);
    }
    for (int i = lane; i < n; i += 64) {
        ta.roots_ad[(size_t)kg * N + i] = a.roots_old[(size_t)anc * N + i];
        ta.cnt_ad[(size_t)kg * N + i] = a.cnt_old[(size_t)anc * N + i];
        ta.rootll_ad[(size_t)kg * N + i] = a.rootll_old[(size_t)anc * N + i];
    }
    const bool local = kg >= a.k0 && kg < a.k0 + a.Kloc;
    if (local && lane == 0) {
        const int k = kg - a.k0;
        a.aux[(size_t)k * PK_AUX + AUX_LL_TILDE] = (a.r > 0) ? a.ll_prev[anc] : a.ll_tilde0;
        if (a.r > 0) a.ancestors[(size_t)(a.r - 1) * a.Kloc + k] = anc;
    }
}

__device__ __forceinline__ void pk_twist_draws_body(const pk_twist_args& ta, const double* __restrict__ Q, int jc, long t) {
    const pk_rank_args& a = ta.a;
    if (t >= 2L * a.Kloc * ta.J) return;
    const int side = (int)(t & 1);
    const long i = t >> 1;
    const int k = (int)(i / ta.J), j = (int)(i - (long)k * ta.J);
    const pm_u32x4 x = pm_philox4x32((uint32_t)(a.k0 + k), (uint32_t)a.r, PM_STREAM_TWIST, (uint32_t)j, a.seed);
    const double b = side ? (-pm_log(pm_unit_oc(x.z, x.w))) / a.lam_r : (-pm_log(pm_unit_oc(x.x, x.y))) / a.lam_l;
    ta.tw_b[i * 2 + side] = b;
    double q[16], p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = Q[u];
    if (jc) pm_jc69(b, p); else pm_expm4(q, b, p);
    double* out = ta.tw_P + i * 32 + side * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) out[u] = p[u];
}
// adoption (one wave per particle) and the draws (independent of it) in one launch: blocks [0, K) adopt, the rest draw
__global__ __launch_bounds__(64) void pk_twist_adopt_draws(const pk_twist_args ta, const double* __restrict__ Q, int jc) {
    if ((int)blockIdx.x < ta.a.K) pk_twist_adopt_body(ta, blockIdx.x);
    else pk_twist_draws_body(ta, Q, jc, (long)(blockIdx.x - ta.a.K) * 64 + threadIdx.x);
}

// one (pair, sub-sample) row of potentials for my canonical column: sites tid, tid + 256, ...
// Two coded leaves: the site likelihood pi . ((leaf_cl P_l) o (leaf_cr P_r)) takes one of 25 values; lik25[cl * 5 + cr] holds
// them, each computed once by exactly the per-site operations (pk_build_lik25), so a site is one lookup: same bits.
__device__ __forceinline__ void pk_build_lik25(const double (*tabL)[4], const double (*tabR)[4], const double* pi4, double* lik25, int t) {
    if (t < 25) {
        const int cl = t / 5, cr = t - cl * 5;
        const double pi[4] = {pi4[0], pi4[1], pi4[2], pi4[3]};
        double o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = tabL[cl][j] * tabR[cr][j];
        lik25[t] = pk_site_lik(pi, o);
    }
}
// one (pair, sub-sample) row of two coded leaves read through both leaf tables (only when the merge reads codes but the
// code-pair histogram is absent): sites [s0, s1) of one tile by one wave, lane = column
__device__ __forceinline__ void pk_twist_row_cc(int s0, int s1, const uint8_t* Lc, const uint8_t* Rc, const double (*tabL)[4],
                                                const double (*tabR)[4], const double (&pi)[4], pm_lp& col) {
    const int lane = threadIdx.x & 63;
    for (int s = s0 + lane; s < s1; s += 64) {
        const int cl = Lc[s], cr = Rc[s];
        double o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = tabL[cl][j] * tabR[cr][j];
        pm_lp_mul(col, pk_site_lik(pi, o));
    }
}

// force a wave-uniform value into scalar registers (the compiler keeps uniform loads in VGPRs once the
// kernel has stored to global memory)
__device__ __forceinline__ double pk_uniform(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
// ---- row loops of the merge that stores nothing (pk_rank_merge_nostore), written for VALU issue, which binds that kernel with
//      ~24 waves per CU in flight (profiles/r02_merge_pmc.json): rows and codes are addressed as scalar base + one 32-bit offset
//      per thread (no 64-bit address arithmetic on the vector pipe, no branch around a load), and two register sets alternate
//      (sites q and q+1 of the thread), so no register-to-register copies rotate the pipeline.
typedef __attribute__((ext_vector_type(4))) unsigned int pk_u4;
typedef __attribute__((address_space(1))) const pk_u4 pk_gu4c;
typedef __attribute__((address_space(1))) const uint8_t pk_gu8c;
struct pk_rowregs { pk_u4 l0, l1, r0, r1; unsigned int cl, cr; };
// wave-uniform base (scalar registers) + one 32-bit byte offset per thread: hipcc then emits the `saddr` form of global_load, with
// no 64-bit address arithmetic on the vector pipe.  The site index is clamped (a site past the end re-reads the last one and its
// factor is replaced by exactly 1.0), so no branch surrounds a load.
__device__ __forceinline__ const char* pk_uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
template <bool CL, bool CR>
__device__ __forceinline__ void pk_rows_load(pk_rowregs& x, const char* bl, const char* br, int s, int s1) {
    const unsigned int sc = (unsigned int)(s < s1 ? s : s1 - 1);
    if constexpr (CL) x.cl = *(pk_gu8c*)(bl + sc);
    else { x.l0 = *(pk_gu4c*)(bl + sc * 32u); x.l1 = *(pk_gu4c*)(bl + sc * 32u + 16u); }
    if constexpr (CR) x.cr = *(pk_gu8c*)(br + sc);
    else { x.r0 = *(pk_gu4c*)(br + sc * 32u); x.r1 = *(pk_gu4c*)(br + sc * 32u + 16u); }
}
__device__ __forceinline__ double pk_u2d(unsigned int lo, unsigned int hi) { return __hiloint2double((int)hi, (int)lo); }
// the new node's row (L P_l) o (R P_r) of the loaded rows / codes
template <bool CL, bool CR>
__device__ __forceinline__ void pk_rows_out(const pk_rowregs& x, const double (&Pl)[16], const double (&Pr)[16], const double (*tabL)[4],
                                            const double (*tabR)[4], double (&o)[4]) {
    double lpv[4], rpv[4];
    if constexpr (CL) {
        const pk_d2 a = *reinterpret_cast<const pk_d2*>(&tabL[x.cl][0]), b = *reinterpret_cast<const pk_d2*>(&tabL[x.cl][2]);
        lpv[0] = a.x; lpv[1] = a.y; lpv[2] = b.x; lpv[3] = b.y;
    } else {
        const double Lv[4] = {pk_u2d(x.l0.x, x.l0.y), pk_u2d(x.l0.z, x.l0.w), pk_u2d(x.l1.x, x.l1.y), pk_u2d(x.l1.z, x.l1.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v = Lv[0] * Pl[j];
            v = pm_fma(Lv[1], Pl[4 + j], v);
            v = pm_fma(Lv[2], Pl[8 + j], v);
            lpv[j] = pm_fma(Lv[3], Pl[12 + j], v);
        }
    }
    if constexpr (CR) {
        const pk_d2 a = *reinterpret_cast<const pk_d2*>(&tabR[x.cr][0]), b = *reinterpret_cast<const pk_d2*>(&tabR[x.cr][2]);
        rpv[0] = a.x; rpv[1] = a.y; rpv[2] = b.x; rpv[3] = b.y;
    } else {
        const double Rv[4] = {pk_u2d(x.r0.x, x.r0.y), pk_u2d(x.r0.z, x.r0.w), pk_u2d(x.r1.x, x.r1.y), pk_u2d(x.r1.z, x.r1.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v = Rv[0] * Pr[j];
            v = pm_fma(Rv[1], Pr[4 + j], v);
            v = pm_fma(Rv[2], Pr[8 + j], v);
            rpv[j] = pm_fma(Rv[3], Pr[12 + j], v);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = lpv[j] * rpv[j];
}
// the site likelihood pi . ((L P_l) o (R P_r)) of the loaded rows / codes
template <bool CL, bool CR>
__device__ __forceinline__ double pk_rows_lik(const pk_rowregs& x, const double (&Pl)[16], const double (&Pr)[16], const double (*tabL)[4],
                                              const double (*tabR)[4], const double* lik25, const double (&pi)[4]) {
    if constexpr (CL && CR) {
        return lik25[x.cl * 5 + x.cr];
    } else {
        double o[4];
        pk_rows_out<CL, CR>(x, Pl, Pr, tabL, tabR, o);
        return pk_site_lik(pi, o);
    }
}
// the loop alone: sites s0 + lane + 64 j < s1 of one tile by one wave (lane = column of the tile), the first register set
// already loaded.  Two sites per trip, ONE renormalisation of the running product for both (pm_lp_mul2: the bits of two
// pm_lp_mul); the trip count is wave-uniform, a lane past the end multiplies by exactly 1.0.
template <bool CL, bool CR>
__device__ __forceinline__ void pk_rows_run(int s0, int s1, const char* bl, const char* br, pk_rowregs& A, const double (&Pl)[16],
                                            const double (&Pr)[16], const double (*tabL)[4], const double (*tabR)[4], const double* lik25,
                                            const double (&pi)[4], pm_lp& col) {
    pk_rowregs B;
    int s = s0 + (int)(threadIdx.x & 63);
    // Measured forms of this loop (20 480-particle launch, primate.p): the one below 37.7 us; both steps
    // computed unconditionally (one padded step when the tile has an odd number of steps, validity selects on both) 41.7;
    // the same with scheduling barriers so that B's rows travel while A is computed 40.8; four register sets, the next pair of
    // steps in flight 39.8 (38.8 at five waves per SIMD).  hipcc sinks B's loads into B's uniform branch and waits for them
    // there, and that is still the fastest: with seven or eight waves per SIMD the launch is bound by instruction issue, not
    // by an exposed load, so the step and the selects that are not executed count for more than the load that is not in flight.
    // Also measured (round 3, 40 960-particle launch): the complete trips peeled off into a loop without validity selects and
    // clamps, both steps unconditional there: 72 against 63 us, single sweep 0.291 against 0.271 ms -- again hipcc interleaves the
    // two steps and waits for both steps' rows at the top.
    #pragma unroll 1
    for (int u = s0; u < s1; u += 128, s += 128) {          // u: wave-uniform
        pk_rows_load<CL, CR>(B, bl, br, s + 64, s1);
        double xa = pk_rows_lik<CL, CR>(A, Pl, Pr, tabL, tabR, lik25, pi);
        xa = s < s1 ? xa : 1.0;
        pk_rows_load<CL, CR>(A, bl, br, s + 128, s1);
        if (u + 64 < s1) {
            double xb = pk_rows_lik<CL, CR>(B, Pl, Pr, tabL, tabR, lik25, pi);
            xb = s + 64 < s1 ? xb : 1.0;
            pm_lp_mul2(col, xa, xb);
        } else {
            pm_lp_mul(col, xa);
        }
    }
}
// one tile of one merge by one wave; the tables are the wave's own slices of LDS
template <bool CL, bool CR>
__device__ __forceinline__ void pk_rows_loop(int s0, int s1, const double* Lp, const double* Rp, const uint8_t* Lc, const uint8_t* Rc,
                                             const double* Pu, const double* pi4, const double (&Pl)[16], const double (&Pr)[16],
                                             double (*tabL)[4], double (*tabR)[4], double* lik25, const double (&pi)[4], pm_lp& col) {
    const int lane = threadIdx.x & 63;
    const char* bl = pk_uniform_ptr(CL ? (const void*)Lc : (const void*)Lp);
    const char* br = pk_uniform_ptr(CR ? (const void*)Rc : (const void*)Rp);
    pk_rowregs A;
    pk_rows_load<CL, CR>(A, bl, br, s0 + lane, s1);        // the first rows / codes travel while the leaf tables are built
    if constexpr (CL || CR) {                              // (the whole wave takes the same variant)
        if (lane < 32) pk_build_leaf_table(Pu, tabL, lane);
        else pk_build_leaf_table(Pu + 16, tabR, lane - 32);
        pk_wave_lds_fence();
    }
    if constexpr (CL && CR) {
        pk_build_lik25(tabL, tabR, pi4, lik25, lane);
        pk_wave_lds_fence();
    }
    pk_rows_run<CL, CR>(s0, s1, bl, br, A, Pl, Pr, tabL, tabR, lik25, pi, col);
}
// contract v4 row (one coded leaf, one internal root X): lik[s] = X[s] . v_code[s], same two-register-set loop
__device__ __forceinline__ void pk_rows_v4(int s0, int s1, const double* Xp, const uint8_t* cd, const double (*vtab)[4], pm_lp& col) {
    const char* bx = pk_uniform_ptr(Xp);
    const char* bc = pk_uniform_ptr(cd);
    pk_u4 a0, a1, b0, b1;
    unsigned int ca, cb;
    auto load = [&](pk_u4& x0, pk_u4& x1, unsigned int& c, int s) {
        const unsigned int sc = (unsigned int)(s < s1 ? s : s1 - 1);
        x0 = *(pk_gu4c*)(bx + sc * 32u); x1 = *(pk_gu4c*)(bx + sc * 32u + 16u);
        c = *(pk_gu8c*)(bc + sc);
    };
    auto site = [&](const pk_u4& x0, const pk_u4& x1, unsigned int c) {
        const pk_d2 va = *reinterpret_cast<const pk_d2*>(&vtab[c][0]), vb = *reinterpret_cast<const pk_d2*>(&vtab[c][2]);
        double lik = pk_u2d(x0.x, x0.y) * va.x;
        lik = pm_fma(pk_u2d(x0.z, x0.w), va.y, lik);
        lik = pm_fma(pk_u2d(x1.x, x1.y), vb.x, lik);
        lik = pm_fma(pk_u2d(x1.z, x1.w), vb.y, lik);
        return lik;
    };
    int s = s0 + (int)(threadIdx.x & 63);
    load(a0, a1, ca, s);
    #pragma unroll 1
    for (int u = s0; u < s1; u += 128, s += 128) {
        load(b0, b1, cb, s + 64);
        double xa = site(a0, a1, ca);
        xa = s < s1 ? xa : 1.0;
        load(a0, a1, ca, s + 128);
        if (u + 64 < s1) {
            double xb = site(b0, b1, cb);
            xb = s + 64 < s1 ? xb : 1.0;
            pm_lp_mul2(col, xa, xb);
        } else {
            pm_lp_mul(col, xa);
        }
    }
}

// The merge of one rank event when the new node is NOT stored (lazy nodes, the last rank event): one WAVE per (particle, tile),
// a workgroup is one wave (two or four independent waves per workgroup measured: 1-3 % slower).  The wave's matrices, pointers
// and site bounds are wave-uniform (scalar registers).  Row-per-lane form: lane c owns column c of the tile, reads whole 32-byte rows (or 1-byte codes), no DPP
// moves: about half the instructions per site of the lane-pair form, whose point is the 16-byte-per-lane store.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void pk_rank_merge_nostore(const pk_rank_args a) {
    __shared__ __attribute__((aligned(16))) double tabL[5][4], tabR[5][4];
    __shared__ double lik25[25];
    const int item = (int)blockIdx.x, ntiles = a.ntiles;
    const int k = ntiles == 1 ? item : item / ntiles, tau = item - k * ntiles;
    const int s0 = tau * a.T, s1 = s0 + a.T < a.S ? s0 + a.T : a.S;
    const double* Pu = a.Pmat + (size_t)k * 32;
    PK_TOUCH_DECL;
    PK_TOUCH_256_2(Pu, a.pi, a.aux + (size_t)k * PK_AUX);        // both matrices, pi, the epilogue's weight terms
    int cl = a.child[k * 2];
    const int cr = a.child[k * 2 + 1];
    PK_TOUCH_END(cl);                                      // (before the first use of the children, on every path)
    const bool codedL = a.leaf_codes && cl < a.N, codedR = a.leaf_codes && cr < a.N;   // wave-uniform
    const double* Lp = pk_node_ptr(a, cl);
    const double* Rp = pk_node_ptr(a, cr);
    const uint8_t* Lc = a.leaf_codes + (codedL ? (size_t)cl * a.S : 0);
    const uint8_t* Rc = a.leaf_codes + (codedR ? (size_t)cr * a.S : 0);
    double Pl[16], Pr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { Pl[u] = Pu[u]; Pr[u] = Pu[16 + u]; }   // uniform address, nothing stored yet: scalar loads
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    pm_lp col = pm_lp_init();
    if (codedL) {
        if (codedR) pk_rows_loop<true, true>(s0, s1, Lp, Rp, Lc, Rc, Pu, a.pi, Pl, Pr, tabL, tabR, lik25, pi, col);
        else pk_rows_loop<true, false>(s0, s1, Lp, Rp, Lc, Rc, Pu, a.pi, Pl, Pr, tabL, tabR, lik25, pi, col);
    } else {
        if (codedR) pk_rows_loop<false, true>(s0, s1, Lp, Rp, Lc, Rc, Pu, a.pi, Pl, Pr, tabL, tabR, lik25, pi, col);
        else pk_rows_loop<false, false>(s0, s1, Lp, Rp, Lc, Rc, Pu, a.pi, Pl, Pr, tabL, tabR, lik25, pi, col);
    }
    const double tot = pk_wave_tree_sum(pm_lp_finish(col));
    if (threadIdx.x == 0) {
        // the epilogue's pointers are read from the kernel-argument segment HERE: taken from `a` they are loaded at the top of the
        // kernel and stay live in scalar registers through the row loops, next to the 64 registers of P_l and P_r
        typedef __attribute__((address_space(4))) const pk_rank_args pk_kernarg;
        pk_kernarg* ka = (pk_kernarg*)__builtin_amdgcn_kernarg_segment_ptr();
        if (ka->ntiles != 1) {
            ka->tilev[(size_t)k * ka->ntiles + tau] = tot;
        } else {
            const int kg = ka->k0 + k;
            const double* ax = ka->aux + (size_t)k * PK_AUX;
            const double fl = ax[AUX_SUM_REM] + tot;
            const double ll = ((fl + ax[AUX_FPRIOR]) + ax[AUX_LPRIOR]) + ax[AUX_RPRIOR];
            const double lw = (((ll - ax[AUX_LL_TILDE]) - ax[AUX_PAREN]) + ax[AUX_LOGV]) - ax[AUX_Q];
            const int N = ka->N, r = ka->r, K = ka->K;
            ka->nodell[N + r * K + kg] = tot;
            ka->rootll_new[(size_t)kg * N + (ka->n - 2)] = tot;
            ka->ll_r[kg] = ll;
            ka->logw_r[kg] = lw;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The Felsenstein merge of one rank event in the form that STORES the node: one workgroup of 256 threads per (local particle,
// site tile); grid = Kloc x ntiles.  Used where every node must be in memory (PHYLO_EAGER_NODES, the twisted proposal, kept
// graphs of long rows, small sharded nodes).
//   k2: the new node's partial likelihoods, out[s,:] = (L[s,:] P_l) * (R[s,:] P_r)      (vcsmc.py:185-187)
//   k3: sum_s log(pi . out[s,:]) in the canonical order                                  (vcsmc.py:240-242)
//   k8: log_likelihood_r and log w_r                                                     (vcsmc.py:376-392)
// Phase 1, all four waves (what should bound this kernel is the store stream): row-per-lane like pk_rank_merge_nostore -- wave w
// takes the 64-site steps w, w + 4, ... of the tile, lane = site, both matrices in scalar registers -- and the 32-byte output
// rows of a step go through 2 KiB of the wave's LDS so that every store instruction writes 1 KiB CONTIGUOUS, 16 bytes per lane
// (non-temporal: only the few particles that survive the next resampling ever read the node again); the site's likelihood
// pi . out goes to LDS.  (Until round 3 a lane PAIR owned a site and the rows were completed with DPP moves: 114 VALU
// instructions per site, a third of them 4-cycle DPP moves, bound the launch at 0.45-0.54 of the HBM peak.  Row-per-lane stores
// of 2 x 16 bytes at a 32-byte stride write half lines and were slower still.)
// Phase 2, wave 0: the tile's 64 column products (contract v5: lane = column, sites in increasing order, read back from LDS),
// one log per lane, the tree, the epilogue.  Same fma chains as everywhere: bit-identical to the other forms.
//   algorithmic traffic: 2 x 32 B read + 32 B written per (particle, site) = 96 B / unit.
// ------------------------------------------------------------------------------------------------
#define PK_MAX_SITE_TILE 4096            // the storing merge keeps a tile's site likelihoods in LDS (8 bytes each)
#define PK_STORE_STAGE_BYTES 8192        // + 2 KiB of output staging per wave
template <bool CL, bool CR, bool STORE>
__device__ __forceinline__ void pk_store_tile(int s0, int s1, int wv, int lane, const double* Lp, const double* Rp, const uint8_t* Lc,
                                              const uint8_t* Rc, double* out, const double (&Pl)[16], const double (&Pr)[16],
                                              const double (*tabL)[4], const double (*tabR)[4], const double (&pi)[4], double* likbuf,
                                              double* stage /*this wave's [64][4]*/) {
    const char* bl = pk_uniform_ptr(CL ? (const void*)Lc : (const void*)Lp);
    const char* br = pk_uniform_ptr(CR ? (const void*)Rc : (const void*)Rp);
    char* ob = const_cast<char*>(pk_uniform_ptr(out));
    pk_rowregs A, B;
    int u = s0 + 64 * wv;                                   // wave-uniform: first site of this wave's step
    if (u >= s1) return;
    pk_rows_load<CL, CR>(A, bl, br, u + lane, s1);
    #pragma unroll 1
    for (; u < s1; u += 256) {
        pk_rows_load<CL, CR>(B, bl, br, u + 256 + lane, s1);        // the next step's rows travel while this one is computed
        double o[4];
        pk_rows_out<CL, CR>(A, Pl, Pr, tabL, tabR, o);
        const int s = u + lane;
        if (s < s1) likbuf[s - s0] = pk_site_lik(pi, o);
        if constexpr (STORE) {
            // transpose through LDS: lane l wrote its row (32 bytes at 32 l); piece i of the step's 2 KiB is read back 16 bytes per
            // lane (offset 1024 i + 16 l) and stored 1 KiB contiguous per instruction
            pk_d2* st = reinterpret_cast<pk_d2*>(stage);
            st[2 * lane] = pk_d2{o[0], o[1]};
            st[2 * lane + 1] = pk_d2{o[2], o[3]};
            pk_wave_lds_fence();
            const int nbytes = (s1 - u < 64 ? s1 - u : 64) * 32;             // valid bytes of this step
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = 1024 * i + 16 * lane;
                const pk_d2 v = st[64 * i + lane];
                if (off < nbytes) __builtin_nontemporal_store(v, (pk_gd2*)(ob + (size_t)u * 32 + off));   // global_store, not flat
            }
            pk_wave_lds_fence();
        }
        A = B;
    }
}

__global__ __launch_bounds__(PK_COLS) __attribute__((amdgpu_waves_per_eu(5, 8))) void pk_rank_merge(const pk_rank_args a) {
    extern __shared__ __attribute__((aligned(16))) double pk_lds_dyn[];     // [tile sites] site likelihoods, then 4 x [64][4] staging
    __shared__ __attribute__((aligned(16))) double tabL[5][4], tabR[5][4];
    const int item = blockIdx.x, k = a.ntiles == 1 ? item : item / a.ntiles, tau = item - k * a.ntiles;
    const int kg = a.k0 + k, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s0 = tau * a.T, s1 = s0 + a.T < a.S ? s0 + a.T : a.S, len = s1 - s0;
    double* likbuf = pk_lds_dyn;
    double* stage = pk_lds_dyn + ((len + 1) & ~1) + (size_t)wv * 256;
    const double* Pu = a.Pmat + (size_t)k * 32;
    PK_TOUCH_DECL;
    PK_TOUCH_256_2(Pu, a.pi, a.aux + (size_t)k * PK_AUX);
    int cl = a.child[k * 2];
    const int cr = a.child[k * 2 + 1];
    PK_TOUCH_END(cl);
    const bool codedL = a.leaf_codes && cl < a.N, codedR = a.leaf_codes && cr < a.N;   // workgroup-uniform
    const double* Lp = pk_node_ptr(a, cl);
    const double* Rp = pk_node_ptr(a, cr);
    const uint8_t* Lc = a.leaf_codes + (codedL ? (size_t)cl * a.S : 0);
    const uint8_t* Rc = a.leaf_codes + (codedR ? (size_t)cr * a.S : 0);
    double* out = a.pool + ((size_t)a.r * a.Kloc + k) * (size_t)a.S * 4;
    double Pl[16], Pr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { Pl[u] = Pu[u]; Pr[u] = Pu[16 + u]; }   // uniform address: scalar loads
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    if (codedL || codedR) {
        if (tid < 32) pk_build_leaf_table(Pu, tabL, tid);
        else if (tid < 64) pk_build_leaf_table(Pu + 16, tabR, tid - 32);
        __syncthreads();
    }
#define PK_MERGE_DISPATCH(ST)                                                                                                          \
    if (codedL) {                                                                                                                      \
        if (codedR) pk_store_tile<true, true, ST>(s0, s1, wv, lane, Lp, Rp, Lc, Rc, out, Pl, Pr, tabL, tabR, pi, likbuf, stage);        \
        else pk_store_tile<true, false, ST>(s0, s1, wv, lane, Lp, Rp, Lc, Rc, out, Pl, Pr, tabL, tabR, pi, likbuf, stage);              \
    } else {                                                                                                                           \
        if (codedR) pk_store_tile<false, true, ST>(s0, s1, wv, lane, Lp, Rp, Lc, Rc, out, Pl, Pr, tabL, tabR, pi, likbuf, stage);       \
        else pk_store_tile<false, false, ST>(s0, s1, wv, lane, Lp, Rp, Lc, Rc, out, Pl, Pr, tabL, tabR, pi, likbuf, stage);             \
    }
    if (a.lazy || a.no_store) { PK_MERGE_DISPATCH(false) } else { PK_MERGE_DISPATCH(true) }
#undef PK_MERGE_DISPATCH
    __syncthreads();
    if (tid >= 64) return;
    // phase 2: lane = column of the tile; the column's sites in increasing order, two per renormalisation
    pm_lp col = pm_lp_init();
    for (int j = tid; j < len; j += 128) {
        const double xa = likbuf[j];
        if (j + 64 < len) pm_lp_mul2(col, xa, likbuf[j + 64]);
        else pm_lp_mul(col, xa);
    }
    const double tot = pk_wave_tree_sum(pm_lp_finish(col));
    if (tid == 0) {
        if (a.ntiles == 1) pk_merge_epilogue(a, k, kg, tot);
        else a.tilev[(size_t)k * a.ntiles + tau] = tot;
    }
}


// (leaf row of `code` . P)[j], bit-identical to pk_build_leaf_table
__device__ __forceinline__ double pk_leaf_entry(const double* __restrict__ P, int code, int j) {
    if (code < 4) return P[code * 4 + j];
    return pm_fma(1.0, P[12 + j], pm_fma(1.0, P[8 + j], pm_fma(1.0, P[4 + j], 1.0 * P[j])));
}

// Potentials of the pairs of two CODED LEAVES: the merged row takes 25 distinct values, so
// sum_s log f(c_l[s], c_r[s]) = sum over code pairs c of count_c log f_c (contract v3): the
// term of code pair c sits in column c of ONE 64-column tile, every other column is 0, same tree.  32 lanes per
// (particle, pair, sub-sample) row; rows of other pairs leave at once (pk_twist_potentials computes those).
__global__ __launch_bounds__(256) void pk_twist_potentials_ll(const pk_twist_args ta) {
    __shared__ __attribute__((aligned(16))) double Psh[8][32];
    __shared__ __attribute__((aligned(16))) double tab[8][2][5][4];
    const pk_rank_args& a = ta.a;
    const int n = a.n, M = ta.M, J = ta.J;
    const int q = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int k = blockIdx.y, j = blockIdx.x * 8 + q, kg = a.k0 + k;   // grid (ceil(J / 8), Kloc)
    const int32_t* ro = ta.roots_ad + (size_t)kg * a.N;
    bool active = j < J;
    int r1 = 0, r2 = 1, idl = 0, idr = 0;
    if (active) {
        int rem = j / M;
        while (rem >= n - 1 - r1) { rem -= n - 1 - r1; ++r1; }
        r2 = r1 + 1 + rem;
        idl = ro[r1]; idr = ro[r2];
        active = idl < a.N && idr < a.N;
    }
    unsigned int cnt = 0;
    if (active) {
        Psh[q][c] = ta.tw_P[((size_t)k * J + j) * 32 + c];
        if (c < 25) cnt = ta.pair_hist[((size_t)idl * a.N + idr) * 32 + c];
    }
    __syncthreads();
    if (active) {
        pk_build_leaf_table(&Psh[q][0], tab[q][0], c);          // 20 entries per side
        if (c < 20) pk_build_leaf_table(&Psh[q][16], tab[q][1], c);
    }
    __syncthreads();
    if (!active) return;
    double term = 0.0;
    if (cnt) {
        const int cl = c / 5, cr = c - cl * 5;
        const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
        double o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = tab[q][0][cl][u] * tab[q][1][cr][u];
        term = (double)cnt * pm_log(pk_site_lik(pi, o));
    }
    double v = 0.0 + term;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v = v + __shfl_xor(v, off, 64);
    if (c == 0) {
        const double tot = v + 0.0;                            // columns 32..63 of the (single) tile's tree are 0
        const int32_t* co = ta.cnt_ad + (size_t)kg * a.N;
        const double* rl = ta.rootll_ad + (size_t)kg * a.N;
        const int c1 = co[r1], c2 = co[r2], c12 = c1 + c2;
        double jp = tot + (-a.ldf[c12 < a.ldf_n ? c12 : a.ldf_n]);
        jp = jp - (rl[r1] + (-a.ldf[c1 < a.ldf_n ? c1 : a.ldf_n]));
        jp = jp - (rl[r2] + (-a.ldf[c2 < a.ldf_n ? c2 : a.ldf_n]));
        ta.pot[(size_t)k * J + j] = jp;
    }
}

// Look-ahead potentials, one WAVE per (particle, pair, sub-sample) row: grid = Kloc * J workgroups of 64 (rounded up to a
// multiple of 8), no workgroup barrier, no staging shared between rows.  (Until round 3 a workgroup owned a (particle, left root)
// and staged eight rows' matrices and tables at a time behind barriers: 76 % of its wave cycles were waits.)  Rows of two coded
// leaves leave at once (pk_twist_potentials_ll prices them by code pair).
//   coded leaf x internal root (contract v4): the wave builds the five vectors v_c in its LDS slice (20 lanes), then X[s] . v_code
//   everything else: the merge's row loops (pk_rows_run) with both matrices in scalar registers
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void pk_twist_potentials(const pk_twist_args ta) {
    __shared__ __attribute__((aligned(16))) double tab[2][5][4];
    __shared__ __attribute__((aligned(16))) double vtab[5][4];
    const pk_rank_args& a = ta.a;
    const int n = a.n, M = ta.M, J = ta.J, lane = threadIdx.x;
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of rows, i.e. of particles, so that the
    // rows of a particle (which read the same internal roots) share one L2.  gridDim.x is a multiple of 8.
    const long nitems = (long)a.Kloc * J;
    const long bid = (long)(blockIdx.x & 7) * (long)(gridDim.x >> 3) + (long)(blockIdx.x >> 3);
    if (bid >= nitems) return;
    const int k = (int)(bid / J), j = (int)(bid - (long)k * J), kg = a.k0 + k;
    int r1 = 0, rem = j / M;                                   // pair t = j / M, lexicographic (r1 < r2)
    while (rem >= n - 1 - r1) { rem -= n - 1 - r1; ++r1; }
    const int r2 = r1 + 1 + rem;
    const int32_t* ro = ta.roots_ad + (size_t)kg * a.N;
    const double* P = ta.tw_P + ((size_t)k * J + j) * 32;      // wave-uniform: scalar loads
    PK_TOUCH_DECL;
    PK_TOUCH_256_2(P, a.pi, ta.cnt_ad + (size_t)kg * a.N);     // the row's matrices, pi, the epilogue's leaf counts: beside the root ids
    int idl = ro[r1];
    const int idr = ro[r2];
    PK_TOUCH_END(idl);                                         // (before the first use of the ids, on every path)
    const bool leafL = idl < a.N, leafR = idr < a.N;
    if (ta.pair_hist && leafL && leafR) return;                // coded leaf x coded leaf: pk_twist_potentials_ll
    const double* Lp = pk_node_ptr(a, idl);
    const double* Rp = pk_node_ptr(a, idr);
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double tot = 0.0;
    if (ta.codes && leafL != leafR) {                          // contract v4
        if (lane < 20) {
            const int c = lane >> 2, ii = lane & 3;
            const double* Pleaf = P + (leafL ? 0 : 16);
            const double* Pint = P + (leafL ? 16 : 0);
            double acc = Pint[ii * 4] * (pi[0] * pk_leaf_entry(Pleaf, c, 0));
            acc = pm_fma(Pint[ii * 4 + 1], pi[1] * pk_leaf_entry(Pleaf, c, 1), acc);
            acc = pm_fma(Pint[ii * 4 + 2], pi[2] * pk_leaf_entry(Pleaf, c, 2), acc);
            vtab[c][ii] = pm_fma(Pint[ii * 4 + 3], pi[3] * pk_leaf_entry(Pleaf, c, 3), acc);
        }
        pk_wave_lds_fence();
        const double* Xp = leafL ? Rp : Lp;
        const uint8_t* cd = ta.codes + (size_t)(leafL ? idl : idr) * a.S;
        for (int s0 = 0; s0 < a.S; s0 += a.T) {
            const int s1 = s0 + a.T < a.S ? s0 + a.T : a.S;
            pm_lp col = pm_lp_init();
            pk_rows_v4(s0, s1, Xp, cd, vtab, col);
            const double t = pk_wave_tree_sum(pm_lp_finish(col));
            tot = s0 ? tot + t : t;
        }
    } else {
        const bool cL = a.leaf_codes && leafL, cR = a.leaf_codes && leafR;
        const uint8_t* Lc = a.leaf_codes + (cL ? (size_t)idl * a.S : 0);
        const uint8_t* Rc = a.leaf_codes + (cR ? (size_t)idr * a.S : 0);
        double Pl[16], Pr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { Pl[u] = P[u]; Pr[u] = P[16 + u]; }
        if (cL || cR) {
            if (lane < 32) pk_build_leaf_table(P, tab[0], lane);
            else pk_build_leaf_table(P + 16, tab[1], lane - 32);
            pk_wave_lds_fence();
        }
        for (int s0 = 0; s0 < a.S; s0 += a.T) {
            const int s1 = s0 + a.T < a.S ? s0 + a.T : a.S;
            pm_lp col = pm_lp_init();
            pk_rowregs A;
            if (cL && cR) {
                pk_twist_row_cc(s0, s1, Lc, Rc, tab[0], tab[1], pi, col);
            } else if (cL) {
                const char* bl = pk_uniform_ptr(Lc); const char* br = pk_uniform_ptr(Rp);
                pk_rows_load<true, false>(A, bl, br, s0 + lane, s1);
                pk_rows_run<true, false>(s0, s1, bl, br, A, Pl, Pr, tab[0], tab[1], nullptr, pi, col);
            } else if (cR) {
                const char* bl = pk_uniform_ptr(Lp); const char* br = pk_uniform_ptr(Rc);
                pk_rows_load<false, true>(A, bl, br, s0 + lane, s1);
                pk_rows_run<false, true>(s0, s1, bl, br, A, Pl, Pr, tab[0], tab[1], nullptr, pi, col);
            } else {
                const char* bl = pk_uniform_ptr(Lp); const char* br = pk_uniform_ptr(Rp);
                pk_rows_load<false, false>(A, bl, br, s0 + lane, s1);
                pk_rows_run<false, false>(s0, s1, bl, br, A, Pl, Pr, tab[0], tab[1], nullptr, pi, col);
            }
            const double t = pk_wave_tree_sum(pm_lp_finish(col));
            tot = s0 ? tot + t : t;
        }
    }
    if (lane == 0) {
        const int32_t* co = ta.cnt_ad + (size_t)kg * a.N;
        const double* rl = ta.rootll_ad + (size_t)kg * a.N;
        const int c1 = co[r1], c2 = co[r2], c12 = c1 + c2;
        double jp = tot + (-a.ldf[c12 < a.ldf_n ? c12 : a.ldf_n]);
        jp = jp - (rl[r1] + (-a.ldf[c1 < a.ldf_n ? c1 : a.ldf_n]));
        jp = jp - (rl[r2] + (-a.ldf[c2 < a.ldf_n ? c2 : a.ldf_n]));
        ta.pot[(size_t)k * J + j] = jp;
    }
}

// one wave per local particle: softmax over its J potentials, one categorical draw (integer CDF), the weight
// terms of vncsmc.py:472-491 for the chosen pair.
// (the body is instantiated twice, with w in LDS and with w in global memory: a pointer selected at run time would turn every
//  access of the common LDS case into a flat access)
__device__ __forceinline__ void pk_twist_choose_body(const pk_twist_args& ta, double* w /*[J]*/) {
    const pk_rank_args& a = ta.a;
    const int k = blockIdx.x, kg = a.k0 + k, lane = threadIdx.x, n = a.n, N = a.N, J = ta.J, M = ta.M;
    const double* pot = ta.pot + (size_t)k * J;
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
    const bool all_bad = !(mx > -pm_inf()) || mx == pm_inf();
    for (int j = lane; j < J; j += 64) {
        const double v = pot[j];
        w[j] = all_bad ? 1.0 : (pm_isnan(v) ? 0.0 : pm_exp(v - mx));
    }
    __syncthreads();
    // integer CDF by the whole wave (integers: any order gives the same total and the same prefix sums): every lane sums the
    // weights of its contiguous chunk, a shuffle scan gives the chunk offsets, the lane whose chunk crosses the threshold
    // finds the index.  Only the floating-point sum of the normaliser keeps its contract order (increasing j, lane 0).
    const int chunk = (J + 63) >> 6, j0 = lane * chunk, j1 = j0 + chunk < J ? j0 + chunk : J;
    unsigned long long mine = 0;
    for (int j = j0; j < j1; ++j) mine += all_bad ? 1ull : (unsigned long long)(w[j] * PM_CDF_SCALE);
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    const uint64_t total = __shfl(incl, 63, 64);
    const pm_u32x4 x = pm_philox4x32((uint32_t)kg, (uint32_t)a.r, PM_STREAM_TWIST, PK_TWIST_DRAW_BLOCK, a.seed);
    const uint64_t thr = pm_mulhi64(((uint64_t)x.y << 32) | x.x, total);
    int cand = J - 1;                                       // first j with (prefix sum through j) > thr, if it lies in my chunk
    const bool crosses = incl > thr && incl - mine <= thr;
    if (crosses) {
        unsigned long long run = incl - mine;
        for (int j = j0; j < j1; ++j) {
            run += all_bad ? 1ull : (unsigned long long)(w[j] * PM_CDF_SCALE);
            if (run > thr) { cand = j; break; }
        }
    }
    const unsigned long long who = __ballot(crosses);
    const int jsel = who ? __shfl(cand, __ffsll((long long)who) - 1, 64) : J - 1;
    // From here on the WHOLE wave works (until round 3 lane 0 walked through some forty dependent loads and thirty-two stores
    // alone: 17 us per rank event for 630 instructions): lane i loads what belongs to root slot i / history entry i / matrix
    // element i, the sums of the contract are then formed in their order from registers (readlane: the indices are uniform),
    // identically on every lane, and the lanes write their own pieces.
    double ssum = 0.0;
    for (int j = 0; j < J; ++j) ssum = ssum + w[j];          // (one address per step: an LDS broadcast)
    const double logq = pot[jsel] - ((all_bad ? 0.0 : mx) + pm_log(ssum));
    const int t = jsel / M;
    int il = 0, rem = t;
    while (rem >= n - 1 - il) { rem -= n - 1 - il; ++il; }
    const int ir = il + 1 + rem;
    const int32_t* ro = ta.roots_ad + (size_t)kg * N;
    const int32_t* co = ta.cnt_ad + (size_t)kg * N;
    const double* rl = ta.rootll_ad + (size_t)kg * N;
    const bool slot = lane < n;                              // (n <= N <= 64)
    const int c_i = slot ? co[lane] : 0, ro_i = slot ? ro[lane] : 0;
    const double rl_i = slot ? rl[lane] : 0.0;
    const double ldf_i = -a.ldf[c_i < a.ldf_n ? c_i : a.ldf_n];
    const double b_l = ta.tw_b[((size_t)k * J + jsel) * 2], b_r = ta.tw_b[((size_t)k * J + jsel) * 2 + 1];
    const bool hist = lane <= a.r;                           // (r <= N - 2)
    const double hl_i = !hist ? 0.0 : (lane == a.r ? b_l : a.bl[(size_t)lane * a.Kloc + k]);
    const double hr_i = !hist ? 0.0 : (lane == a.r ? b_r : a.br[(size_t)lane * a.Kloc + k]);
    const double P_i = lane < 32 ? ta.tw_P[((size_t)k * J + jsel) * 32 + lane] : 0.0;
    double sum_rem = 0.0, fprior = 0.0;
    int vminus = 0;
    for (int i = n - 1; i >= 0; --i) {                       // remaining roots, descending slot order
        if (i == il || i == ir) continue;
        const int c = __builtin_amdgcn_readlane(c_i, i);
        sum_rem = sum_rem + pk_readlane(rl_i, i);
        fprior = fprior + pk_readlane(ldf_i, i);
        vminus += c - (c == 1 ? 1 : 0);
    }
    const int cnew = __builtin_amdgcn_readlane(c_i, il) + __builtin_amdgcn_readlane(c_i, ir);
    fprior = fprior + (-a.ldf[cnew < a.ldf_n ? cnew : a.ldf_n]);
    vminus += cnew - (cnew == 1 ? 1 : 0);
    double lp = 0.0, rp = 0.0;
    for (int j = 0; j <= a.r; ++j) {
        lp = lp + ((-a.lam_l) * pk_readlane(hl_i, j) + a.loglam_l);
        rp = rp + ((-a.lam_r) * pk_readlane(hr_i, j) + a.loglam_r);
    }
    if (lane < 32) ta.Pmat_r[(size_t)k * 32 + lane] = P_i;
    if (lane == 0) {
        ta.chosen[kg] = (double)jsel;
        ta.bl_r[k] = b_l;
        ta.br_r[k] = b_r;
        double* ax = a.aux + (size_t)k * PK_AUX;
        ax[AUX_SUM_REM] = sum_rem;
        ax[AUX_FPRIOR] = fprior;
        ax[AUX_LPRIOR] = lp;
        ax[AUX_RPRIOR] = rp;
        ax[AUX_PAREN] = ((a.loglam_l - a.lam_l * b_l) + a.loglam_r) - a.lam_r * b_r;
        ax[AUX_LOGV] = pm_log((double)vminus);
        ax[AUX_Q] = logq;                                    // vncsmc.py:491 subtracts the normalised log-potential
        a.child[k * 2] = __builtin_amdgcn_readlane(ro_i, il);
        a.child[k * 2 + 1] = __builtin_amdgcn_readlane(ro_i, ir);
        a.merges[((size_t)a.r * a.Kloc + k) * 2] = il;
        a.merges[((size_t)a.r * a.Kloc + k) * 2 + 1] = ir;
    }
    if (ta.own_tables) {                                     // one GPU: my own new root table (pk_twist_tables otherwise)
        if (slot) {
            const bool merged = lane == il || lane == ir;
            const int p = (n - 1 - lane) - (il > lane ? 1 : 0) - (ir > lane ? 1 : 0);   // remaining roots in descending slot order
            if (!merged) {
                a.roots_new[(size_t)kg * N + p] = ro_i;
                a.cnt_new[(size_t)kg * N + p] = c_i;
                a.rootll_new[(size_t)kg * N + p] = rl_i;
            }
            if (a.pos_hist) a.pos_hist[(size_t)kg * N + lane] = merged ? -1 : p;
        }
        if (lane == 0) {
            a.roots_new[(size_t)kg * N + (n - 2)] = N + a.r * a.K + kg;
            a.cnt_new[(size_t)kg * N + (n - 2)] = cnew;
        }
    }
}

__global__ __launch_bounds__(64) void pk_twist_choose(const pk_twist_args ta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ta.J <= PK_TWIST_LDS_J) pk_twist_choose_body(ta, reinterpret_cast<double*>(smem));
    else pk_twist_choose_body(ta, ta.wbuf + (size_t)blockIdx.x * ta.J);
}

__global__ void pk_twist_tables(const pk_twist_args ta) {
    const pk_rank_args& a = ta.a;
    const int kg = blockIdx.x * blockDim.x + threadIdx.x;
    if (kg >= a.K) return;
    const int n = a.n, N = a.N, t = (int)ta.chosen[kg] / ta.M;
    int il = 0, rem = t;
    while (rem >= n - 1 - il) { rem -= n - 1 - il; ++il; }
    const int ir = il + 1 + rem;
    const int32_t* ro = ta.roots_ad + (size_t)kg * N;
    const int32_t* co = ta.cnt_ad + (size_t)kg * N;
    const double* rl = ta.rootll_ad + (size_t)kg * N;
    int p = 0;
    for (int i = n - 1; i >= 0; --i) {
        if (i == il || i == ir) continue;
        a.roots_new[(size_t)kg * N + p] = ro[i];
        a.cnt_new[(size_t)kg * N + p] = co[i];
        a.rootll_new[(size_t)kg * N + p] = rl[i];
        if (a.pos_hist) a.pos_hist[(size_t)kg * N + i] = p;
        ++p;
    }
    if (a.pos_hist) { a.pos_hist[(size_t)kg * N + il] = -1; a.pos_hist[(size_t)kg * N + ir] = -1; }
    a.roots_new[(size_t)kg * N + p] = N + a.r * a.K + kg;
    a.cnt_new[(size_t)kg * N + p] = co[il] + co[ir];
}

// ================================================================================================
// Multi-GPU, one process per GPU: the exchange of a rank event WITHOUT a collective call ("a one-shot P2P
// write into peers' buffers + flag"; the RCCL all-gather of phylo_comm.h stays as the compare / fallback path, PHYLO_P2P=0).
// Every rank owns an EXCHANGE SLAB (fine-grained device memory, the same layout on every rank, mapped by every peer through
// hipIpc like the node pools): the K-vectors that are all-gathered (log-weights, log-likelihoods, node log-likelihoods, the
// twisted proposal's choices) and one 64-bit flag per peer and purpose.  One workgroup per rank:
//   1. copies THIS rank's segments of up to four arrays into every peer's slab over xGMI (system-scope write-through stores,
//      consecutive lanes -> consecutive addresses), fence, workgroup barrier;
//   2. stores the exchange's epoch (monotone per context; every rank issues the same exchanges in the same order) into its flag
//      in every peer's slab (release, system scope);
//   3. lanes poll the peers' flags in the OWN slab until they reach the epoch (bounded by wall-clock time: ten seconds of the
//      100 MHz real-time counter unless PHYLO_P2P_WAIT_S says otherwise -- ranks may be apart by host work --, then the timeout word is set and phylo_sweep_fetch reports the sweep as invalid), one
//      system-scope acquire, barrier.
// A rank can be at most one exchange ahead of a peer (it needs the peer's flag of exchange e to finish e), and an array's row
// is written again only N - 1 >= 2 exchanges later, so no row is overwritten before its readers are done (contexts of two
// taxa, one exchange per sweep, use the collective).  With n_seg = 0 the kernel is a barrier (the "owners have written their
// adopted nodes" barrier of lazy nodes).  No host call, no second stream, no communicator shared between contexts.
// ================================================================================================
#define PK_P2P_WAIT_TICKS 1000000000ull      // default bound of a flag wait: 10 s of s_memrealtime (100 MHz); PHYLO_P2P_WAIT_S
struct pk_p2p_args {
    char* const* slabs;                      // [world] every rank's exchange slab as mapped in this process (own slab at [me])
    int world, me;
    int n_seg;                               // arrays to exchange (0: barrier only)
    size_t seg_off[4];                       // byte offset of each array in the slab; rank p's segment starts at + p * seg_count doubles
    int seg_count;                           // doubles per rank and array
    size_t flag_off;                         // byte offset of this purpose's flags[world] (u64) in the slab
    unsigned long long epoch;
    unsigned long long wait_ticks;           // bound of the flag wait in 100 MHz ticks
    unsigned int* timeout_word;
};
// the copy alone, over many workgroups (large exchanges: batched sweeps move 3 x 8 x Kloc bytes to every peer; one workgroup's
// store issue would take longer than the merge): the kernel boundary behind it orders the writes before the flags that
// pk_p2p_exchange (with n_seg = 0) then raises
__global__ __launch_bounds__(1024) void pk_p2p_copy(const pk_p2p_args a) {
    const long per = (long)a.n_seg * a.seg_count, total = per * (a.world - 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int p = (int)(i / per);
        const long j = i - (long)p * per;
        p += p >= a.me ? 1 : 0;                                  // the peers in rank order, this rank left out
        const int seg = (int)(j / a.seg_count), e = (int)(j - (long)seg * a.seg_count);
        const size_t at = a.seg_off[seg] + ((size_t)a.me * a.seg_count + e) * 8;
        const double v = *reinterpret_cast<const double*>(a.slabs[a.me] + at);
        __hip_atomic_store(reinterpret_cast<double*>(a.slabs[p] + at), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(1024) void pk_p2p_exchange(const pk_p2p_args a) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = a.n_seg * a.seg_count;
    for (int p = 0; p < a.world; ++p) {
        if (p == a.me) continue;
        for (int i = tid; i < per; i += nt) {
            const int seg = i / a.seg_count, e = i - seg * a.seg_count;
            const size_t at = a.seg_off[seg] + ((size_t)a.me * a.seg_count + e) * 8;
            const double v = *reinterpret_cast<const double*>(a.slabs[a.me] + at);      // written by this rank's previous kernel
            __hip_atomic_store(reinterpret_cast<double*>(a.slabs[p] + at), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope
    __syncthreads();
    if (tid < a.world && tid != a.me)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.slabs[tid] + a.flag_off) + a.me, a.epoch, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < a.world && tid != a.me) {
        const unsigned long long* f = reinterpret_cast<const unsigned long long*>(a.slabs[a.me] + a.flag_off) + tid;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.wait_ticks) {
                __hip_atomic_store(a.timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");            // system scope
    __syncthreads();
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
        case 4: v = pm_exp_nonpos(x[i]); break;
        default: v = pm_fma(x[i], y[i], x[i]); break;
    }
    out[i] = v;
}
