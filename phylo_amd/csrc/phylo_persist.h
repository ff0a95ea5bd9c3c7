// phylo_persist.h -- the whole CSMC sweep (vcsmc.py:406-451: every rank event of body_rank_update, :332-400) as ONE
// launch on one GPU: plain proposal, lazy nodes, G >= 1 independent sweeps ("groups").  DESIGN.md section 4c.
//
// Why: issued as launches, a rank event is scan -> bookkeeping -> materialise -> merge, four DEPENDENT launches that
// each sit at their latency floor (5-10 us) with the GPU mostly idle, 0.34 ms per K = 2048 sweep; the reference's own
// loop (tf.while_loop, vcsmc.py:430-438; one evaluation sweep per epoch, :529-538) cannot batch sweeps to hide that.
//
// Structure.  Group g (one sweep of Kg particles) is run by Wg resident workgroups of 256 threads; workgroup `wl` of the
// group owns the m = Kg / Wg particles [wl m, (wl + 1) m) for the whole sweep.  The only all-to-all dependency of a rank
// event is weights -> cdf; it is crossed ONCE per rank event by an arrival counter:
//   every workgroup publishes the log-weights of its particles (write-through stores, every wave drained, workgroup
//   barrier, one agent-scope add on the group's monotone counter), then polls the counter, acquires, and reads ALL Kg
//   log-weights of its group (16 KiB at Kg = 2048, L2-served) -- the "allgather" hand-off of MI355X_MICROARCH.md's price
//   list, recipe R1 of cdna_hip_programming.md Guideline 16.
// Every workgroup then runs the SAME deterministic scan (max, canonical sum, integer cdf; pk_scan_block) redundantly with
// the cdf in LDS: no second hop for a broadcast, and the index searches hit LDS.  The resampling uniforms of ALL particles
// (Philox, independent of the state) are drawn once in the prologue and shared through memory, so each workgroup can
// price every draw of its group against the cdf interval of ITS OWN particles: it learns which of its nodes of the
// previous rank event were adopted (lazy nodes: only those are ever read again) and writes exactly those into the pool
// before its own merges; a merge that needs a node written during the same rank event waits on that node's mark.
// Owners never wait while they write, so the waits cannot form a cycle; every spin is bounded and reports a timeout.
// Merges: one WAVE per particle (no workgroup barrier inside a merge): lane l owns column l of every site tile (contract v5:
// sites tile start + l + 64 j), the column tree is one in-wave butterfly per tile.  Arithmetic, orders and outputs are those of the
// launch path (phylo_kernels.h) bit for bit; tests compare both with the C oracle.
#pragma once
#include "phylo_kernels.h"

#define PP_CHUNK 16            // particles of a workgroup whose bookkeeping lives in LDS at one time
#define PP_CTR_STRIDE 16       // u64 words between the arrival counters of two groups (one 128-byte line each)
#define PP_MAX_M 1024          // particles per workgroup (adoption flags and own ancestors live in LDS)
#define PP_MAX_KG 8192         // particles per group (the group's cdf lives in LDS: 8 bytes each)
#define PP_SPIN_LIMIT (1u << 21)

struct pp_args {
    int N, S, K, Kg, G, R;           // K = G * Kg particles, R = N - 1 rank events
    int T;                           // contract v5: sites per tile of the canonical sum over sites
    int Wg, m;                       // workgroups per group, particles per workgroup (Kg = Wg * m); grid = G * Wg
    uint64_t seed;                   // G == 1
    const uint64_t* group_seeds;     // [G] or NULL
    uint32_t flags;
    int jc;
    const double* Q; const double* lam_l; const double* lam_r; const double* pi; const double* ldf;
    const double* leaves; const uint8_t* leaf_codes; double* pool;
    int32_t* roots[2]; int32_t* cnt[2]; double* rootll[2];       // [K][N], double-buffered over rank events
    double* nodell; double* bl; double* br; double* Pmat; double* logw; double* ll;
    int32_t* child; int32_t* merges; int64_t* anc; unsigned int* mark;
    unsigned long long* rdraw;       // [R][K] resampling draws (row 0 unused)
    double* lse; int lse_stride;     // [G][R + 1]
    unsigned long long* ctr;         // [G][PP_CTR_STRIDE] monotone arrival counters
    unsigned long long ctr_base;     // their value before this launch
    unsigned int* timeout_word;
    unsigned long long* stamps;      // NULL, or [R + 1][PP_NSTAMP] s_memrealtime ticks (100 MHz) of workgroup 0 (PHYLO_PERSIST_STAMPS=1)
};
#define PP_NSTAMP 16
// phase stamps of workgroup 0, thread 0 (diagnostic runs only: the pointer is NULL otherwise)
__device__ __forceinline__ void pp_stamp(const pp_args& a, int r, int i) {
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[(size_t)r * PP_NSTAMP + i] = __builtin_amdgcn_s_memrealtime();
}

// per-particle LDS slot: what part A (before the wait) leaves for part B + merge (after it); one wave owns a slot
struct pp_slot {
    double *aux, *P, *tab, *lik25;   // aux[PK_AUX]; P[32] = {P_l, P_r}; tab[2][5][4] leaf lookup tables built from P; lik25: see pk_build_lik25
    uint32_t* key; int32_t *rank, *inv, *misc;   // key[n4]; rank[slot] (-1: merged); inv[rank] = slot; misc[0] = il, [1] = ir
};
__host__ __device__ inline size_t pp_slot_bytes(int N) {
    const size_t n4 = ((size_t)N + 3) & ~(size_t)3;
    return (PK_AUX + 32 + 40 + 26) * 8 + (3 * n4 + 4) * 4;
}
__device__ __forceinline__ pp_slot pp_carve(char* base, int N) {
    const size_t n4 = ((size_t)N + 3) & ~(size_t)3;
    pp_slot L;
    L.aux = (double*)base;
    L.P = L.aux + PK_AUX;
    L.tab = L.P + 32;
    L.lik25 = L.tab + 40;
    L.key = (uint32_t*)(L.lik25 + 26);
    L.rank = (int32_t*)(L.key + n4);
    L.inv = L.rank + n4;
    L.misc = L.inv + n4;
    return L;
}
struct pp_scan_lds { double d[16]; unsigned long long u[16]; };
// LDS of one workgroup: scan scratch | ldf table | flags | own ancestors | adoption flags | particle slots | cdf
struct pp_lds_layout { size_t ldf, flags, anc, adopted, slots, cdf, total; };
__host__ __device__ inline pp_lds_layout pp_layout(int N, int m, int Kg) {
    pp_lds_layout o;
    size_t p = (sizeof(pp_scan_lds) + 15) & ~(size_t)15;
    o.ldf = p; p += (((size_t)N + 2) & ~(size_t)1) * 8;
    p = (p + 15) & ~(size_t)15;
    o.flags = p; p += 16;
    o.anc = p; p += (((size_t)m + 3) & ~(size_t)3) * 4;
    o.adopted = p; p += (((size_t)m + 3) & ~(size_t)3) * 4;
    p = (p + 15) & ~(size_t)15;
    o.slots = p; p += (size_t)PP_CHUNK * ((pp_slot_bytes(N) + 15) & ~(size_t)15);
    o.cdf = p; p += (size_t)Kg * 8;
    o.total = p;
    return o;
}

typedef __attribute__((address_space(1))) const double pp_gdc;
typedef __attribute__((address_space(1))) const unsigned long long pp_gu64c;
typedef __attribute__((address_space(1))) const int32_t pp_gi32c;
__device__ __forceinline__ double pp_gld(const double* p) { return *(pp_gdc*)p; }
__device__ __forceinline__ unsigned long long pp_gld(const unsigned long long* p) { return *(pp_gu64c*)p; }
__device__ __forceinline__ int32_t pp_gld(const int32_t* p) { return *(pp_gi32c*)p; }
__device__ __forceinline__ void pp_st_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double pp_readlane(double v, int l) {      // l wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// The kernel's hot code must stay inside the instruction cache (64 KiB shared by two CUs; every rank event walks scan, search,
// bookkeeping and a merge once): exp and log are CALLED, not inlined at their two dozen sites, and the merge keeps a rolled
// loop.  Same operations as pm_exp / pm_log / pm_lp_mul / pm_lp_finish, same bits.
__device__ __attribute__((noinline)) double pp_log(double x) { return pm_log(x); }
__device__ __attribute__((noinline)) double pp_exp(double x) { return pm_exp(x); }
__device__ __forceinline__ void pp_lp_mul(pm_lp& a, double x) {
    const uint64_t bx = pm_bits(x);
    const int ex = (int)((bx >> 52) & 0x7ff);
    if (bx - 0x0010000000000000ull >= 0x7fe0000000000000ull) {        // not a positive normal number: rare, out of line
        a.extra = a.extra + pp_log(x);
        return;
    }
    const double mx = pm_from_bits((bx & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    a.p = a.p * mx;
    const uint64_t bp = pm_bits(a.p);
    a.E += (ex - 1023) + ((int)((bp >> 52) & 0x7ff) - 1023);
    a.p = pm_from_bits((bp & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
}
// the common path of pm_lp_mul alone (x a positive normal number): no branch, so hipcc can interleave site steps
__device__ __forceinline__ void pp_lp_mul_fast(pm_lp& a, double x) {
    const uint64_t bx = pm_bits(x);
    const int ex = (int)((bx >> 52) & 0x7ff);
    const double mx = pm_from_bits((bx & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    a.p = a.p * mx;
    const uint64_t bp = pm_bits(a.p);
    a.E += (ex - 1023) + ((int)((bp >> 52) & 0x7ff) - 1023);
    a.p = pm_from_bits((bp & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
}
__device__ __forceinline__ bool pp_lp_special(double x) { return pm_bits(x) - 0x0010000000000000ull >= 0x7fe0000000000000ull; }
__device__ __forceinline__ double pp_lp_finish(const pm_lp& a) {
    const double dE = (double)a.E;
    return ((pp_log(a.p) + dE * 1.90821492927058770002e-10) + dE * 6.93147180369123816490e-01) + a.extra;
}

// Canonical sum over the 64 columns held by the lanes of one wave (the adjacent-pair tree of pk_block_canon_sum: xor 1, 2,
// 4, 8, 16, 32), returned wave-uniform.  Levels 1..8 are DPP moves inside a row of 16 lanes (after levels 1 and 2 a quad holds
// one value, so the mirrored partner of level 4 / 8 holds exactly the value of the xor partner); the four row totals are
// then read by lane and added as (r0 + r1) + (r2 + r3).  a + b is commutative bit for bit: same result as the butterfly,
// without its six dependent trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double pp_dpp(double v) { return pk_dpp<CTRL>(v); }
__device__ __forceinline__ double pp_wave_tree_sum(double v) { return pk_wave_tree_sum(v); }

// wave maximum by the same DPP steps (max is exact and commutative), wave-uniform result
__device__ __forceinline__ double pp_wave_max(double v) {
    double o;
    o = pp_dpp<0xB1>(v); v = o > v ? o : v;
    o = pp_dpp<0x4E>(v); v = o > v ? o : v;
    o = pp_dpp<0x141>(v); v = o > v ? o : v;
    o = pp_dpp<0x140>(v); v = o > v ? o : v;
    const double r0 = pp_readlane(v, 0), r1 = pp_readlane(v, 16), r2 = pp_readlane(v, 32), r3 = pp_readlane(v, 48);
    const double a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}
// inclusive prefix sum of one u64 per lane over the wave (integers: any association gives the same bits): row_shr 1, 2, 4,
// 8 inside rows of 16 (lanes shifted in read 0), then the three row totals are added by lane reads
template <int CTRL>
__device__ __forceinline__ unsigned long long pp_dpp_shr_u64(unsigned long long v) {
    int lo = (int)(unsigned int)v, hi = (int)(unsigned int)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ unsigned long long pp_readlane_u64(unsigned long long v, int l) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long pp_wave_incl_scan_u64(unsigned long long v, int lane) {
    v += pp_dpp_shr_u64<0x111>(v);    // row_shr:1
    v += pp_dpp_shr_u64<0x112>(v);    // row_shr:2
    v += pp_dpp_shr_u64<0x114>(v);    // row_shr:4
    v += pp_dpp_shr_u64<0x118>(v);    // row_shr:8
    const unsigned long long t0 = pp_readlane_u64(v, 15), t1 = pp_readlane_u64(v, 31), t2 = pp_readlane_u64(v, 47);
    const int row = lane >> 4;
    return v + (row > 0 ? t0 : 0ull) + (row > 1 ? t1 : 0ull) + (row > 2 ? t2 : 0ull);
}

// arrival: every handed-off byte was stored write-through (sc1); every wave drains, the workgroup meets, ONE lane adds
__device__ __forceinline__ void pp_arrive(unsigned long long* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: ONE lane polls relaxed (bounded), ONE agent acquire, the workgroup barrier holds everybody for the invalidate.
// Returns false when the poll gave up (the workgroup then leaves the kernel: its tables may be incomplete).
__device__ __forceinline__ bool pp_wait(unsigned long long* ctr, unsigned long long target, unsigned int* timeout_word, int* lds_flag) {
    if (threadIdx.x == 0) {
        unsigned int spins = 0;
        int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PP_SPIN_LIMIT) {
                __hip_atomic_store(timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// ---- k5 in LDS, by NT threads: pk_scan_block's arithmetic (max; w = exp(logw - max); canonical sum over 256 columns;
//      integer weights floor(w 2^44) and their inclusive prefix sum) with the cdf left in LDS.  The doubles w are staged in
//      cdf[] itself, so the column sums (thread c < 256 adds w[c], w[c + 256], ... in that order, whatever NT is) and the
//      integer conversion read them from LDS.  logw was published by other workgroups: read after the acquire of pp_wait.
template <int NT>
__device__ __forceinline__ void pp_scan(const double* logw, int Kg, unsigned long long* cdf, double* lse_out, pp_scan_lds* sh,
                                        bool want_cdf) {
    constexpr int E = 2048 / NT, NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool one_tile = Kg <= 2048;
    double v[E];
    double m = -pm_inf();
    for (int base = 0; base < Kg; base += 2048) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int k = base + tid + NT * j;
            v[j] = pp_gld(logw + (k < Kg ? k : Kg - 1));
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (base + tid + NT * j >= Kg) v[j] = pm_nan();
            if (!pm_isnan(v[j]) && v[j] > m) m = v[j];
        }
    }
    m = pp_wave_max(m);
    if (lane == 0) sh->d[wv] = m;
    __syncthreads();
    m = sh->d[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = sh->d[i] > m ? sh->d[i] : m;
    const bool all_bad = !(m > -pm_inf()) || m == pm_inf();
    for (int base = 0; base < Kg; base += 2048) {
        if (!one_tile) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int k = base + tid + NT * j;
                v[j] = pp_gld(logw + (k < Kg ? k : Kg - 1));
            }
        }
        // w = exp(logw - max), straight-line (pm_exp_nonpos == pm_exp for arguments <= 0): the E evaluations overlap
        double w[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool nan = pm_isnan(v[j]);
            const double e = pm_exp_nonpos(nan ? 0.0 : v[j] - m);
            w[j] = all_bad ? 1.0 : (nan ? 0.0 : e);
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int k = base + tid + NT * j;
            if (k < Kg) cdf[k] = pm_bits(w[j]);
        }
    }
    __syncthreads();
    if (tid < 256) {                                   // canonical fp sum: column c = k mod 256, increasing k
        double col = 0.0;
        for (int k = tid; k < Kg; k += 256) col = col + pm_from_bits(cdf[k]);
        col = pp_wave_tree_sum(col);
        if (lane == 0) sh->d[wv] = col;
    }
    __syncthreads();
    if (tid == 0 && lse_out) {
        const double sum = ((sh->d[0] + sh->d[1]) + sh->d[2]) + sh->d[3];
        const double mm = all_bad ? 0.0 : m;
        *lse_out = (mm + pp_log(sum)) - pp_log((double)Kg);
    }
    if (!want_cdf) return;
    unsigned long long carry = 0;
    for (int base = 0; base < Kg; base += 2048) {
        const int lo = base + E * tid;
        unsigned long long e[E];
#pragma unroll
        for (int j = 0; j < E; ++j)
            e[j] = (lo + j < Kg) ? (all_bad ? 1ull : (unsigned long long)(pm_from_bits(cdf[lo + j]) * PM_CDF_SCALE)) : 0ull;
#pragma unroll
        for (int j = 1; j < E; ++j) e[j] += e[j - 1];
        const unsigned long long local = e[E - 1];
        const unsigned long long incl = pp_wave_incl_scan_u64(local, lane);
        if (base > 0) __syncthreads();                 // the previous tile's wave totals have been read
        if (lane == 63) sh->u[wv] = incl;
        __syncthreads();
        unsigned long long run = carry + incl - local, all = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (i < wv) run += sh->u[i];
            all += sh->u[i];
        }
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (lo + j < Kg) cdf[lo + j] = run + e[j];
        carry += all;
    }
}

// The same scan as a launch of its own (launches-per-rank-event path, phylo_resample, phylo_log_zsmc): one workgroup of 512
// threads per group, cdf built in LDS and copied out.  Replaces pk_resample_scan(_groups) whenever the group fits LDS.
// logz_R > 0: lse_out is element logz_R - 1 of a row of logz_R log-normalisers; their left-to-right sum (compute_log_ZSMC,
// vcsmc.py:276) goes to element logz_R of the row -- the last scan of a sweep then needs no pk_logz_total launch behind it.
template <int NT>
__global__ __launch_bounds__(NT) void pp_resample_scan(const double* logw, int Kg, uint64_t* __restrict__ cdf, double* lse_out, int lse_stride,
                                                       int logz_R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pp_scan_lds* sh = reinterpret_cast<pp_scan_lds*>(smem);
    unsigned long long* lcdf = reinterpret_cast<unsigned long long*>(smem + ((sizeof(pp_scan_lds) + 15) & ~(size_t)15));
    const int g = blockIdx.x;
    pp_scan<NT>(logw + (size_t)g * Kg, Kg, lcdf, lse_out ? lse_out + (size_t)g * lse_stride : (double*)nullptr, sh, cdf != nullptr);
    if (logz_R > 0 && lse_out && threadIdx.x == 0) {       // thread 0 wrote the last element itself, the others come from earlier launches
        double* row = lse_out + (size_t)g * lse_stride - (logz_R - 1);
        double z = 0.0;
        for (int r = 0; r < logz_R; ++r) z = z + row[r];
        row[logz_R] = z;
    }
    if (!cdf) return;
    __syncthreads();
    unsigned long long* out = reinterpret_cast<unsigned long long*>(cdf) + (size_t)g * Kg;
    for (int k = threadIdx.x; k < Kg; k += NT) out[k] = lcdf[k];
}
// ---- The same scan by SEVERAL workgroups per group (large groups: the replicated scan of a sharded sweep -- 8 GPUs x 2048 particles
//      is one group of 16 384 weights, 27 us for the one workgroup above, as long as the merge beside it -- and any K beyond the
//      LDS form).  Three small launches, a tile of 2048 weights per workgroup:
//        pp_scan_multi_max   the tile's maximum (exact, any order);
//        pp_scan_multi_exp   the group's maximum from the tiles', w = exp(logw - max) stored as bits, the tile's sum of the INTEGER
//                            weights floor(w 2^44) (exact, any order);
//        pp_scan_multi_cdf   tile b: the integer prefix sum behind the tiles before it; one more workgroup per group: the canonical
//                            floating-point sum (column c = k mod 256, increasing k, then the tree -- the order of pp_scan, which
//                            one workgroup has to walk) and the log-normaliser.
//      Bit for bit pp_scan's results: only integer sums and maxima are re-associated.
#define PP_SCAN_TILE 2048
struct pp_scan_multi_args {
    const double* logw;                // [G][Kg]
    int Kg, B;                         // B = tiles per group
    double* gmax;                      // [G][B]
    unsigned long long* bsum;          // [G][B]
    unsigned long long* wbits;         // [G][Kg]
    unsigned long long* cdf;           // [G][Kg] or nullptr
    double* lse_out;                   // or nullptr
    int lse_stride, logz_R;
    int lse_only;                      // pp_scan_multi_cdf: no cdf wanted, the launch is the log-normaliser's workgroup alone
};
__device__ __forceinline__ double pp_scan_multi_group_max(const pp_scan_multi_args& a, int g) {
    double m = pp_gld(a.gmax + (size_t)g * a.B);
    for (int b = 1; b < a.B; ++b) {
        const double o = pp_gld(a.gmax + (size_t)g * a.B + b);
        m = o > m ? o : m;
    }
    return m;
}
__global__ __launch_bounds__(512) void pp_scan_multi_max(const pp_scan_multi_args a) {
    __shared__ double sh[8];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* logw = a.logw + (size_t)g * a.Kg;
    double m = -pm_inf();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = b * PP_SCAN_TILE + tid + 512 * j;
        const double v = pp_gld(logw + (k < a.Kg ? k : a.Kg - 1));
        if (k < a.Kg && !pm_isnan(v) && v > m) m = v;
    }
    m = pp_wave_max(m);
    if (lane == 0) sh[wv] = m;
    __syncthreads();
    if (tid == 0) {
        m = sh[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = sh[i] > m ? sh[i] : m;
        a.gmax[(size_t)g * a.B + b] = m;
    }
}
__global__ __launch_bounds__(512) void pp_scan_multi_exp(const pp_scan_multi_args a) {
    __shared__ unsigned long long sh[8];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* logw = a.logw + (size_t)g * a.Kg;
    const double m = pp_scan_multi_group_max(a, g);
    const bool all_bad = !(m > -pm_inf()) || m == pm_inf();
    double v[4], w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = b * PP_SCAN_TILE + tid + 512 * j;
        v[j] = pp_gld(logw + (k < a.Kg ? k : a.Kg - 1));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // (straight-line, like pp_scan: the four evaluations overlap)
        const bool nan = pm_isnan(v[j]);
        const double e = pm_exp_nonpos(nan ? 0.0 : v[j] - m);
        w[j] = all_bad ? 1.0 : (nan ? 0.0 : e);
    }
    unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = b * PP_SCAN_TILE + tid + 512 * j;
        if (k < a.Kg) {
            a.wbits[(size_t)g * a.Kg + k] = pm_bits(w[j]);
            local += all_bad ? 1ull : (unsigned long long)(w[j] * PM_CDF_SCALE);
        }
    }
    const unsigned long long incl = pp_wave_incl_scan_u64(local, lane);
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += sh[i];
        a.bsum[(size_t)g * a.B + b] = t;
    }
}
__global__ __launch_bounds__(512) void pp_scan_multi_cdf(const pp_scan_multi_args a) {
    __shared__ pp_scan_lds sh;
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long* wb = a.wbits + (size_t)g * a.Kg;
    const double m = pp_scan_multi_group_max(a, g);
    const bool all_bad = !(m > -pm_inf()) || m == pm_inf();
    if (b == a.B || a.lse_only) {                          // the canonical floating-point sum and the log-normaliser
        if (!a.lse_out) return;
        if (tid < 256) {                                   // (sixteen loads in flight, the additions in the order of pp_scan)
            double col = 0.0;
            for (int k0 = tid; k0 < a.Kg; k0 += 256 * 16) {
                unsigned long long q[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = k0 + 256 * u;
                    q[u] = pp_gld(wb + (k < a.Kg ? k : a.Kg - 1));
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (k0 + 256 * u < a.Kg) col = col + pm_from_bits(q[u]);
            }
            col = pp_wave_tree_sum(col);
            if (lane == 0) sh.d[wv] = col;
        }
        __syncthreads();
        if (tid == 0) {
            const double sum = ((sh.d[0] + sh.d[1]) + sh.d[2]) + sh.d[3];
            const double mm = all_bad ? 0.0 : m;
            double* out = a.lse_out + (size_t)g * a.lse_stride;
            *out = (mm + pp_log(sum)) - pp_log((double)a.Kg);
            if (a.logz_R > 0) {                            // (the earlier elements of the row come from earlier launches)
                double* row = out - (a.logz_R - 1);
                double z = 0.0;
                for (int r = 0; r < a.logz_R; ++r) z = z + row[r];
                row[a.logz_R] = z;
            }
        }
        return;
    }
    unsigned long long carry = 0;
    for (int i = 0; i < b; ++i) carry += pp_gld(a.bsum + (size_t)g * a.B + i);
    const int lo = b * PP_SCAN_TILE + 4 * tid;
    unsigned long long e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        e[j] = (lo + j < a.Kg) ? (all_bad ? 1ull : (unsigned long long)(pm_from_bits(pp_gld(wb + lo + j)) * PM_CDF_SCALE)) : 0ull;
#pragma unroll
    for (int j = 1; j < 4; ++j) e[j] += e[j - 1];
    const unsigned long long local = e[3];
    const unsigned long long incl = pp_wave_incl_scan_u64(local, lane);
    if (lane == 63) sh.u[wv] = incl;
    __syncthreads();
    unsigned long long run = carry + incl - local;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < wv) run += sh.u[i];
    unsigned long long* out = a.cdf + (size_t)g * a.Kg;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (lo + j < a.Kg) out[lo + j] = run + e[j];
}
#define PP_SCAN_KERNEL_MAX_KG 16384    // 128 KiB of cdf in LDS, 1024 threads: the replicated scan of 8 GPUs x 2048 particles
__host__ inline size_t pp_resample_scan_lds(int Kg) { return ((sizeof(pp_scan_lds) + 15) & ~(size_t)15) + (size_t)Kg * 8; }

// ---- part A of a particle's rank event, by ONE wave, BEFORE the wait (nothing here depends on the resampling): the pair
//      pick of extend_partial_state (vcsmc.py:303-305: Gumbel top-2 restated on the keys, remaining slots by ascending key),
//      the branch-history priors with this rank's rate (quirk Q3, vcsmc.py:378-384), the proposal terms, and the particle's
//      transition matrices + leaf lookup tables staged in LDS.  Same arithmetic and orders as pk_book_packed.
template <bool STAGE>      // STAGE: also stage the particle's transition matrices, leaf tables and lik25 in the slot (one-launch sweep)
__device__ __forceinline__ void pp_part_a(const pp_args& a, int r, int kg, uint32_t kin, uint64_t seed, const pp_slot& L, int lane,
                                          double lam_l, double lam_r, double loglam_l, double loglam_r) {
    const int N = a.N, n = N - r, K = a.K;
    const int nb = (n + 3) / 4;
    pp_stamp(a, r, 13);
    if (lane < nb) {
        const pm_u32x4 x = pm_philox4x32(kin, (uint32_t)r, PM_STREAM_PAIR, (uint32_t)lane, seed);
        L.key[lane * 4 + 0] = x.x; L.key[lane * 4 + 1] = x.y; L.key[lane * 4 + 2] = x.z; L.key[lane * 4 + 3] = x.w;
    }
    const double hbl = lane <= r ? pp_gld(a.bl + (size_t)lane * K + kg) : 0.0;     // rows 0..r of my branch history
    const double hbr = lane <= r ? pp_gld(a.br + (size_t)lane * K + kg) : 0.0;
    if constexpr (STAGE) { if (lane < 32) L.P[lane] = pp_gld(a.Pmat + ((size_t)r * K + kg) * 32 + lane); }
    pp_lds_fence();
    // largest key (lower slot on ties), then the second largest
    unsigned long long best = lane < n ? (((unsigned long long)L.key[lane] << 32) | (0xffffffffu - (uint32_t)lane)) : 0ull;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    const int il = (int)(0xffffffffu - (uint32_t)best);
    best = (lane < n && lane != il) ? (((unsigned long long)L.key[lane] << 32) | (0xffffffffu - (uint32_t)lane)) : 0ull;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    const int ir = (int)(0xffffffffu - (uint32_t)best);
    // position of every remaining slot in ascending (key, slot) order
    if (lane < n) {
        int rank = -1;
        if (lane != il && lane != ir) {
            const unsigned long long mine = ((unsigned long long)L.key[lane] << 32) | (uint32_t)lane;
            rank = 0;
            #pragma unroll 1
            for (int j = 0; j < n; ++j) {
                const unsigned long long cj = ((unsigned long long)L.key[j] << 32) | (uint32_t)j;
                rank += (j != il && j != ir && cj < mine) ? 1 : 0;
            }
            L.inv[rank] = lane;
        }
        L.rank[lane] = rank;
    }
    if constexpr (STAGE) {
        if (lane < 20) pk_build_leaf_table(L.P, reinterpret_cast<double (*)[4]>(L.tab), lane);
        else if (lane >= 32 && lane < 52) pk_build_leaf_table(L.P + 16, reinterpret_cast<double (*)[4]>(L.tab + 20), lane - 32);
        pp_lds_fence();
        pk_build_lik25(reinterpret_cast<const double (*)[4]>(L.tab), reinterpret_cast<const double (*)[4]>(L.tab + 20), a.pi, L.lik25, lane);
    }
    double lp = 0.0, rp = 0.0;                            // history rows 0..r with THIS rank's rate (quirk Q3)
    #pragma unroll 1
    for (int j = 0; j <= r; ++j) {
        lp = lp + ((-lam_l) * pp_readlane(hbl, j) + loglam_l);
        rp = rp + ((-lam_r) * pp_readlane(hbr, j) + loglam_r);
    }
    const double b_l = pp_readlane(hbl, r), b_r = pp_readlane(hbr, r);
    if (lane == 0) {
        const double q = 1.0 / ((double)((n - 1) * n) / 2.0);          // 1 / ncr(n, 2), vcsmc.py:298
        L.aux[AUX_LPRIOR] = lp;
        L.aux[AUX_RPRIOR] = rp;
        L.aux[AUX_PAREN] = ((loglam_l - lam_l * b_l) + loglam_r) - lam_r * b_r;
        L.aux[AUX_Q] = (a.flags & 1u) ? q : pp_log(q);
        L.misc[0] = il;
        L.misc[1] = ir;
    }
    pp_lds_fence();
    pp_stamp(a, r, 14);
}

// ---- the merge of one particle by ONE wave, tile after tile: lane l owns column l of the tile, i.e. sites s0 + l + 64 j.
//      With one wave per SIMD nothing else hides a memory round trip, so the rows / codes of the NEXT BS site steps are
//      in flight while BS are computed (two register sets, ping-pong; the loop stays rolled: instruction cache).
template <bool CL, bool CR, int BS>
struct pp_blk {
    pk_d2 La[CL ? 1 : BS][2], Ra[CR ? 1 : BS][2];
    int cl[CL ? BS : 1], cr[CR ? BS : 1];
};
template <bool CL, bool CR, int BS>
__device__ __forceinline__ void pp_blk_load(pp_blk<CL, CR, BS>& b, int s0, int s1, int it0, const double* Lp, const double* Rp, const uint8_t* Lc,
                                            const uint8_t* Rc, int lane) {
    // no branch around a load (hipcc then counts its waits exactly and keeps the next block in flight): a site step past
    // the end re-reads the last site; its factor is replaced by 1.0 below
#pragma unroll
    for (int u = 0; u < BS; ++u) {
        int s = s0 + lane + 64 * (it0 + u);
        s = s < s1 ? s : s1 - 1;
        if constexpr (CL) b.cl[u] = Lc[s]; else { b.La[u][0] = pk_gload2(Lp + (size_t)s * 4); b.La[u][1] = pk_gload2(Lp + (size_t)s * 4 + 2); }
        if constexpr (CR) b.cr[u] = Rc[s]; else { b.Ra[u][0] = pk_gload2(Rp + (size_t)s * 4); b.Ra[u][1] = pk_gload2(Rp + (size_t)s * 4 + 2); }
    }
}
template <bool CL, bool CR, int BS>
__device__ __forceinline__ void pp_blk_compute(const pp_blk<CL, CR, BS>& b, int s0, int s1, int it0, const double (&Pl)[16], const double (&Pr)[16],
                                               const double (*tabL)[4], const double (*tabR)[4], const double* lik25,
                                               const double (&pi)[4], pm_lp& col, bool& special, int lane) {
#pragma unroll
    for (int u = 0; u < BS; ++u) {
        const int s = s0 + lane + 64 * (it0 + u);
        if constexpr (CL && CR) {                       // two coded leaves: one of 25 memoised site likelihoods
            double lik = lik25[b.cl[u] * 5 + b.cr[u]];
            lik = s < s1 ? lik : 1.0;
            const bool sp = pp_lp_special(lik);
            special |= sp;
            pp_lp_mul_fast(col, sp ? 1.0 : lik);
        } else {
            double lpv[4], rpv[4], o[4];
            if constexpr (CL) {
                const pk_d2 x = *reinterpret_cast<const pk_d2*>(&tabL[b.cl[u]][0]), y = *reinterpret_cast<const pk_d2*>(&tabL[b.cl[u]][2]);
                lpv[0] = x.x; lpv[1] = x.y; lpv[2] = y.x; lpv[3] = y.y;
            } else {
                const double Lv[4] = {b.La[u][0].x, b.La[u][0].y, b.La[u][1].x, b.La[u][1].y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double v = Lv[0] * Pl[j];
                    v = pm_fma(Lv[1], Pl[4 + j], v);
                    v = pm_fma(Lv[2], Pl[8 + j], v);
                    lpv[j] = pm_fma(Lv[3], Pl[12 + j], v);
                }
            }
            if constexpr (CR) {
                const pk_d2 x = *reinterpret_cast<const pk_d2*>(&tabR[b.cr[u]][0]), y = *reinterpret_cast<const pk_d2*>(&tabR[b.cr[u]][2]);
                rpv[0] = x.x; rpv[1] = x.y; rpv[2] = y.x; rpv[3] = y.y;
            } else {
                const double Rv[4] = {b.Ra[u][0].x, b.Ra[u][0].y, b.Ra[u][1].x, b.Ra[u][1].y};
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
            // a factor of exactly 1.0 leaves the running product's mantissa and exponent bit for bit unchanged.  A factor that is
            // not a positive normal number (pm_lp_mul's rare branch) is only flagged here: the caller redoes the particle out of line
            double lik = pk_site_lik(pi, o);
            lik = s < s1 ? lik : 1.0;
            const bool sp = pp_lp_special(lik);
            special |= sp;
            pp_lp_mul_fast(col, sp ? 1.0 : lik);
        }
    }
}
// sites [s0, s1) of one tile
template <bool CL, bool CR, int BS>
__device__ __forceinline__ bool pp_merge_wave(int s0, int s1, const double* Lp, const double* Rp, const uint8_t* Lc, const uint8_t* Rc,
                                              const double (&Pl)[16], const double (&Pr)[16], const double (*tabL)[4],
                                              const double (*tabR)[4], const double* lik25, const double (&pi)[4], pm_lp& col,
                                              int lane) {
    const int nit = (s1 - s0 + 63) >> 6;
    bool special = false;
    pp_blk<CL, CR, BS> A, B;
    pp_blk_load<CL, CR, BS>(A, s0, s1, 0, Lp, Rp, Lc, Rc, lane);
    #pragma unroll 1
    for (int it0 = 0; it0 < nit; it0 += 2 * BS) {
        pp_blk_load<CL, CR, BS>(B, s0, s1, it0 + BS, Lp, Rp, Lc, Rc, lane);
        pp_blk_compute<CL, CR, BS>(A, s0, s1, it0, Pl, Pr, tabL, tabR, lik25, pi, col, special, lane);
        pp_blk_load<CL, CR, BS>(A, s0, s1, it0 + 2 * BS, Lp, Rp, Lc, Rc, lane);
        pp_blk_compute<CL, CR, BS>(B, s0, s1, it0 + BS, Pl, Pr, tabL, tabR, lik25, pi, col, special, lane);
    }
    return special;
}

// The same merge in its plainest form (rolled, pm_lp_mul with its rare branch), out of line: run for a particle in which some
// site likelihood was not a positive normal number (zero, subnormal, negative, inf, NaN).  Leaves the row's sum in g[0].
__device__ __attribute__((noinline)) void pp_merge_slow(int S, int T, const double* Lp, const double* Rp, const double* P /*32, LDS*/,
                                                       const double* pi4, double* g /*LDS*/) {
    const int lane = threadIdx.x & 63;
    double Pl[16], Pr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { Pl[u] = P[u]; Pr[u] = P[16 + u]; }
    const double pi[4] = {pi4[0], pi4[1], pi4[2], pi4[3]};
    double tot = 0.0;
    #pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += T) {
        const int s1 = s0 + T < S ? s0 + T : S;
        pm_lp col = pm_lp_init();
        #pragma unroll 1
        for (int s = s0 + lane; s < s1; s += 64) {
            double Lv[4], Rv[4], o[4];
            pk_load4(Lp + (size_t)s * 4, Lv);
            pk_load4(Rp + (size_t)s * 4, Rv);
            pk_merge_site(Lv, Rv, Pl, Pr, o);
            pm_lp_mul(col, pk_site_lik(pi, o));
        }
        const double t = pp_wave_tree_sum(pm_lp_finish(col));
        tot = s0 ? tot + t : t;
    }
    if (lane == 0) g[0] = tot;
}

__device__ __forceinline__ const double* pp_node_ptr(const pp_args& a, int id) {
    const size_t node_sz = (size_t)a.S * 4;
    return id < a.N ? a.leaves + (size_t)id * node_sz : a.pool + (size_t)(id - a.N) * node_sz;
}

// ---- part B + merge of one particle, by ONE wave, after the resampling: adoption of the ancestor's root table (the
//      tf.gather of vcsmc.py:286-288 on integer tables), the new table (:361-373), the weight terms that depend on it
//      (:376-392), the Felsenstein merge of the picked pair (:180-188) with its log-likelihood (:240-242), log w (:392).
struct pp_b_out { int cl, cr; double sum_rem, fprior, logv, ll_tilde; };
// WT: the next plane of the root tables is stored write-through (other workgroups of the SAME launch adopt it at the next
// rank event: one-launch sweep); plain stores when a kernel boundary follows (two-launch rank event).
template <bool WT>
__device__ __forceinline__ pp_b_out pp_part_b(const pp_args& a, int r, int kg, int gbase, int anc, int il, int ir, const int32_t* inv,
                                              const double* ldf, double ll_tilde0, int lane) {
    const int N = a.N, n = N - r, K = a.K, cur = r & 1, nxt = cur ^ 1;
    // my lane's slot of the ancestor's table
    const bool in = lane < n;
    const int node = in ? pp_gld(a.roots[cur] + (size_t)anc * N + lane) : 0;
    const int c = in ? pp_gld(a.cnt[cur] + (size_t)anc * N + lane) : 0;
    const double xll = in ? pp_gld(a.rootll[cur] + (size_t)anc * N + lane) : 0.0;
    pp_b_out o;
    o.ll_tilde = (r > 0) ? pp_gld(a.ll + (size_t)(r - 1) * K + anc) : ll_tilde0;
    o.cl = __builtin_amdgcn_readlane(node, il);
    o.cr = __builtin_amdgcn_readlane(node, ir);
    const int cnew = __builtin_amdgcn_readlane(c, il) + __builtin_amdgcn_readlane(c, ir);
    // the new table in position order: lane q < n-2 takes the slot whose rank is q, lane n-2 the node created now
    const int src = lane < n - 2 ? inv[lane] : 0;
    int onode = __shfl(node, src, 64), oc = __shfl(c, src, 64);
    const double oll = __shfl(xll, src, 64);
    if (lane == n - 2) { onode = N + r * K + kg; oc = cnew; }
    if (lane < n - 1) {
        if constexpr (WT) {
            pp_st_i32(a.roots[nxt] + (size_t)kg * N + lane, onode);
            pp_st_i32(a.cnt[nxt] + (size_t)kg * N + lane, oc);
            if (lane < n - 2) pk_st_agent(a.rootll[nxt] + (size_t)kg * N + lane, oll);
        } else {
            a.roots[nxt][(size_t)kg * N + lane] = onode;
            a.cnt[nxt][(size_t)kg * N + lane] = oc;
            if (lane < n - 2) a.rootll[nxt][(size_t)kg * N + lane] = oll;
        }
    }
    const double oldf = lane < n - 1 ? ldf[oc < N ? oc : N] : 0.0;
    if (lane == 0) {
        a.merges[((size_t)r * K + kg) * 2 + 0] = il;
        a.merges[((size_t)r * K + kg) * 2 + 1] = ir;
        a.child[((size_t)r * K + kg) * 2 + 0] = o.cl;
        a.child[((size_t)r * K + kg) * 2 + 1] = o.cr;
        if (r > 0) a.anc[(size_t)(r - 1) * K + kg] = anc - gbase;        // index inside the group
    }
    double sum_rem = 0.0, fprior = 0.0;                    // sequential sums in position order (values are wave-uniform)
    int vminus = 0;
    #pragma unroll 1
    for (int p = 0; p < n - 2; ++p) sum_rem = sum_rem + pp_readlane(oll, p);
    #pragma unroll 1
    for (int p = 0; p < n - 1; ++p) {
        const int cc = __builtin_amdgcn_readlane(oc, p);
        fprior = fprior + (-pp_readlane(oldf, p));
        vminus += cc - (cc == 1 ? 1 : 0);
    }
    o.sum_rem = sum_rem;
    o.fprior = fprior;
    o.logv = pp_log((double)vminus);
    return o;
}

template <int BS>
__device__ __forceinline__ void pp_part_b_merge(const pp_args& a, int r, int kg, int gbase, int anc, const pp_slot& L, const double* ldf,
                                                const double (&pi)[4], double ll_tilde0, int lane) {
    const int N = a.N, n = N - r, K = a.K, S = a.S, nxt = (r & 1) ^ 1;
    pp_stamp(a, r, 8);
    const int il = __builtin_amdgcn_readfirstlane(L.misc[0]), ir = __builtin_amdgcn_readfirstlane(L.misc[1]);
    const pp_b_out b = pp_part_b<true>(a, r, kg, gbase, anc, il, ir, L.inv, ldf, ll_tilde0, lane);
    const int cl = b.cl, cr = b.cr;
    const double sum_rem = b.sum_rem, fprior = b.fprior, logv = b.logv, ll_tilde = b.ll_tilde;
    pp_stamp(a, r, 9);
    // a child created at the previous rank event is being written by its owner during THIS rank event
    const int fresh0 = N + (r - 1) * K;
    if (r > 0 && (cl >= fresh0 || cr >= fresh0)) {
        unsigned int spins = 0;
        for (;;) {
            const unsigned int ml = cl >= fresh0 ? __hip_atomic_load(a.mark + (cl - N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            const unsigned int mr = cr >= fresh0 ? __hip_atomic_load(a.mark + (cr - N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            if (ml && mr) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PP_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(a.timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const bool codedL = a.leaf_codes && cl < N, codedR = a.leaf_codes && cr < N;
    const double* Lp = pp_node_ptr(a, cl);
    const double* Rp = pp_node_ptr(a, cr);
    const uint8_t* Lc = a.leaf_codes + (codedL ? (size_t)cl * S : 0);
    const uint8_t* Rc = a.leaf_codes + (codedR ? (size_t)cr * S : 0);
    double Pl[16], Pr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { Pl[u] = L.P[u]; Pr[u] = L.P[16 + u]; }
    pp_stamp(a, r, 10);
    const double (*tabL)[4] = reinterpret_cast<const double (*)[4]>(L.tab);
    const double (*tabR)[4] = reinterpret_cast<const double (*)[4]>(L.tab + 20);
    bool special = false;
    double tot = 0.0;
    #pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += a.T) {                  // tiles left to right (contract v5); primate.p: one tile
        const int s1 = s0 + a.T < S ? s0 + a.T : S;
        pm_lp col = pm_lp_init();
        if (codedL) {
            if (codedR) special |= pp_merge_wave<true, true, BS>(s0, s1, Lp, Rp, Lc, Rc, Pl, Pr, tabL, tabR, L.lik25, pi, col, lane);
            else special |= pp_merge_wave<true, false, BS>(s0, s1, Lp, Rp, Lc, Rc, Pl, Pr, tabL, tabR, L.lik25, pi, col, lane);
        } else {
            if (codedR) special |= pp_merge_wave<false, true, BS>(s0, s1, Lp, Rp, Lc, Rc, Pl, Pr, tabL, tabR, L.lik25, pi, col, lane);
            else special |= pp_merge_wave<false, false, BS>(s0, s1, Lp, Rp, Lc, Rc, Pl, Pr, tabL, tabR, L.lik25, pi, col, lane);
        }
        const double t = pp_wave_tree_sum(pp_lp_finish(col));
        tot = s0 ? tot + t : t;
    }
    if (__any(special)) {                                  // rare: redo the particle with the contract's rare branch in place
        pp_merge_slow(S, a.T, Lp, Rp, L.P, a.pi, L.tab);   // (rows are read directly: a leaf row times P equals its table entry bit
        pp_lds_fence();                                    //  for bit; the particle's tables are no longer needed: scratch for the sum)
        tot = L.tab[0];
        pp_lds_fence();
    }
    pp_stamp(a, r, 11);
    if (lane == 0) {                                      // k8: log_likelihood_r and log w_r (vcsmc.py:376-392)
        const double fl = sum_rem + tot;
        const double ll = ((fl + fprior) + L.aux[AUX_LPRIOR]) + L.aux[AUX_RPRIOR];
        const double lw = (((ll - ll_tilde) - L.aux[AUX_PAREN]) + logv) - L.aux[AUX_Q];
        a.nodell[N + r * K + kg] = tot;
        pk_st_agent(a.rootll[nxt] + (size_t)kg * N + (n - 2), tot);
        pk_st_agent(a.ll + (size_t)r * K + kg, ll);
        pk_st_agent(a.logw + (size_t)r * K + kg, lw);
    }
    pp_lds_fence();                                       // the slot is rewritten by part A of the next rank event
    pp_stamp(a, r, 12);
}

template <int NT>
__global__ __launch_bounds__(NT, NT / 256) void pp_sweep(const pp_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = a.N, S = a.S, K = a.K, Kg = a.Kg, R = a.R, m = a.m;
    const int g = blockIdx.x % a.G, wl = blockIdx.x / a.G;      // blocks b, b + 8 share an XCD: a group of G = 8 k sweeps stays on few L2s
    const int gbase = g * Kg, kbl = wl * m, kb = gbase + kbl;   // my particles: global kb .. kb + m - 1, in-group kbl ..
    const uint64_t seed = a.group_seeds ? a.group_seeds[g] : a.seed;
    const pp_lds_layout lay = pp_layout(N, m, Kg);
    pp_scan_lds* sh = reinterpret_cast<pp_scan_lds*>(smem);
    double* ldf = reinterpret_cast<double*>(smem + lay.ldf);
    int* flags = reinterpret_cast<int*>(smem + lay.flags);        // [0] wait ok, [1] some node of mine was adopted
    int* anc_own = reinterpret_cast<int*>(smem + lay.anc);
    int* adopted = reinterpret_cast<int*>(smem + lay.adopted);
    char* slots = smem + lay.slots;
    const size_t slot_stride = (pp_slot_bytes(N) + 15) & ~(size_t)15;
    unsigned long long* cdf = reinterpret_cast<unsigned long long*>(smem + lay.cdf);
    unsigned long long* ctr = a.ctr + (size_t)g * PP_CTR_STRIDE;
    const double ll_tilde0 = pp_log(1.0 / (double)Kg);           // vcsmc.py:422

    // ---- prologue: everything that does not depend on the particle state, for MY particles and every rank event
    pp_stamp(a, R, 0);
    if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[(size_t)R * PP_NSTAMP + 4] = __builtin_amdgcn_s_memtime();
    for (int t = tid; t <= N; t += NT) ldf[t] = a.ldf[t];
    {   // branch lengths b = -log(U)/lambda_r and their transition matrices (vcsmc.py:351-356, 181-184): pk_sweep_draws
        double q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] = a.Q[j];
        for (int t = tid; t < 2 * R * m; t += NT) {
            const int side = t & 1, i = t >> 1, r = i / m, j = i - r * m, kg = kb + j;
            const pm_u32x4 x = pm_philox4x32((uint32_t)(kbl + j), (uint32_t)r, PM_STREAM_BRANCH, 0u, seed);
            const double b = side ? (-pm_log(pm_unit_oc(x.z, x.w))) / a.lam_r[r] : (-pm_log(pm_unit_oc(x.x, x.y))) / a.lam_l[r];
            (side ? a.br : a.bl)[(size_t)r * K + kg] = b;
            double p[16];
            if (a.jc) pm_jc69(b, p); else pm_expm4(q, b, p);
            double* out = a.Pmat + ((size_t)r * K + kg) * 32 + side * 16;
#pragma unroll
            for (int u = 0; u < 16; ++u) out[u] = p[u];
        }
    }
    for (int t = tid; t < (R - 1) * m; t += NT) {                  // resampling draws of rank events 1..R-1 (vcsmc.py:285)
        const int r = 1 + t / m, j = t - (r - 1) * m;
        const pm_u32x4 x = pm_philox4x32((uint32_t)(kbl + j), (uint32_t)r, PM_STREAM_RESAMPLE, 0u, seed);
        __hip_atomic_store(a.rdraw + (size_t)r * K + kb + j, ((unsigned long long)x.y << 32) | x.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = tid; t < R * m; t += NT) {                        // no node of mine is in the pool yet
        const int r = t / m, j = t - r * m;
        __hip_atomic_store(a.mark + (size_t)r * K + kb + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = tid; t < m * N; t += NT) {                        // root tables before the first rank event: the leaves
        const int j = t / N, i = t - j * N;
        a.roots[0][(size_t)(kb + j) * N + i] = i;
        a.cnt[0][(size_t)(kb + j) * N + i] = 1;
        a.rootll[0][(size_t)(kb + j) * N + i] = a.nodell[i];
    }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    pp_stamp(a, R, 1);
    constexpr int BS = NT == 256 ? 4 : 2;
    // One copy of every phase (the hot loop must fit the instruction cache).  Step r < R is rank event r; step R only closes
    // the sweep (the last log-normaliser and the sum, by one workgroup of the group).
    for (int r = 0; r <= R; ++r) {
        pp_stamp(a, r < R ? r : R - 1, r < R ? 0 : 15);
        if (r == R && wl != 0) break;
        const double lam_l = a.lam_l[r < R ? r : 0], lam_r = a.lam_r[r < R ? r : 0];
        const double loglam_l = pp_log(lam_l), loglam_r = pp_log(lam_r);
        for (int c0 = 0; c0 < m || c0 == 0; c0 += PP_CHUNK) {
            const int cnt = m - c0 < PP_CHUNK ? m - c0 : PP_CHUNK;
            // part A of this chunk's particles; for the first chunk this runs BEFORE the wait, while the other workgroups
            // are still finishing the previous rank event
            if (r < R)
                for (int p = wv; p < cnt; p += NW)
                    pp_part_a<true>(a, r, kb + c0 + p, (uint32_t)(kbl + c0 + p), seed, pp_carve(slots + (size_t)p * slot_stride, N), lane,
                              lam_l, lam_r, loglam_l, loglam_r);
            if (c0 == 0 && r > 0) {
                // ---- every workgroup of the group has published log w_{r-1}: one hop
                if (!pp_wait(ctr, a.ctr_base + (unsigned long long)r * a.Wg, a.timeout_word, flags)) return;
                pp_stamp(a, r < R ? r : R - 1, r < R ? 1 : 15);
                for (int t = tid; t < m; t += NT) adopted[t] = 0;
                if (tid == 0) flags[1] = 0;
                // the SAME scan in every workgroup, cdf left in LDS; one workgroup keeps the log-normaliser
                pp_scan<NT>(a.logw + (size_t)(r - 1) * K + gbase, Kg, cdf,
                            wl == 0 ? a.lse + (size_t)g * a.lse_stride + (r - 1) : (double*)nullptr, sh, r < R);
                if (r == R) break;
                __syncthreads();
                pp_stamp(a, r, 2);
                // which of MY nodes of rank event r-1 were adopted: price every draw of the group against my cdf interval
                const unsigned long long total = cdf[Kg - 1];
                const unsigned long long lo = kbl ? cdf[kbl - 1] : 0ull, hi = cdf[kbl + m - 1];
                const unsigned long long* rd = a.rdraw + (size_t)r * K + gbase;
                if (hi > lo) {
                    for (int j0 = 0; j0 < Kg; j0 += 8 * NT) {
                        unsigned long long Rj[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int j = j0 + tid + NT * i;
                            Rj[i] = j < Kg ? pp_gld(rd + j) : 0ull;
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int j = j0 + tid + NT * i;
                            const unsigned long long thr = pm_mulhi64(Rj[i], total);
                            if (j < Kg && thr >= lo && thr < hi) {
                                const int idx = pk_cdf_search(reinterpret_cast<const uint64_t*>(cdf + kbl), m, thr);
                                adopted[idx] = 1;
                                flags[1] = 1;
                            }
                        }
                    }
                }
                for (int t = tid; t < m; t += NT)                    // the ancestors of MY particles (vcsmc.py:285)
                    anc_own[t] = pk_cdf_search(reinterpret_cast<const uint64_t*>(cdf), Kg, pm_mulhi64(pp_gld(rd + kbl + t), total));
                __syncthreads();
                pp_stamp(a, r, 3);
                // ---- lazy nodes: write my adopted nodes of rank event r-1 (children: leaves or nodes written at earlier rank events)
                if (flags[1]) {
                    for (int j = 0; j < m; ++j) {
                        if (!adopted[j]) continue;
                        const size_t x = (size_t)(r - 1) * K + kb + j;
                        const int32_t* ch = a.child + x * 2;
                        const double* Lp = pp_node_ptr(a, ch[0]);
                        const double* Rp = pp_node_ptr(a, ch[1]);
                        const double* P = a.Pmat + x * 32;
                        double Pl[16], Pr[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { Pl[u] = P[u]; Pr[u] = P[16 + u]; }
                        double* out = a.pool + x * (size_t)S * 4;
                        for (int s = tid; s < S; s += NT) {
                            double Lv[4], Rv[4], o[4];
                            pk_load4(Lp + (size_t)s * 4, Lv);
                            pk_load4(Rp + (size_t)s * 4, Rv);
                            pk_merge_site(Lv, Rv, Pl, Pr, o);
#pragma unroll
                            for (int u = 0; u < 4; ++u) pk_st_agent(out + (size_t)s * 4 + u, o[u]);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    for (int t = tid; t < m; t += NT)
                        if (adopted[t]) __hip_atomic_store(a.mark + (size_t)(r - 1) * K + kb + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                pp_stamp(a, r, 4);
            }
            // ---- part B + merge of this chunk's particles: one wave each
            for (int p = wv; p < cnt; p += NW) {
                const int j = c0 + p;
                pp_part_b_merge<BS>(a, r, kb + j, gbase, r > 0 ? gbase + anc_own[j] : kb + j, pp_carve(slots + (size_t)p * slot_stride, N), ldf, pi,
                                    ll_tilde0, lane);
            }
        }
        if (r == R) break;
        pp_stamp(a, r, 6);
        pp_arrive(ctr);
        pp_stamp(a, r, 7);
    }
    if (a.stamps && blockIdx.x == 0 && tid == 0) {
        a.stamps[(size_t)R * PP_NSTAMP + 5] = __builtin_amdgcn_s_memtime();
        a.stamps[(size_t)R * PP_NSTAMP + 6] = __builtin_amdgcn_s_memrealtime();
    }
    // ---- log Z-hat (vcsmc.py:270-277, 445-447): the sum of the log-normalisers over rank events
    if (wl != 0) return;
    if (tid == 0) {
        double* lse = a.lse + (size_t)g * a.lse_stride;
        double z = 0.0;
        for (int r = 0; r < R; ++r) z = z + lse[r];
        lse[R] = z;
    }
}

