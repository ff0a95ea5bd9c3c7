// merge_bench.hip -- stand-alone ablation harness for the Felsenstein merge kernel on MI355X.
// Not part of the library: a measuring tool.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//   tools/merge_bench.hip -o /tmp/merge_bench ; run on the GPU box.
// Shapes follow primate.p K=2048: N=12 leaves, S=898 sites, 11 output slabs of K nodes each.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../phylo_amd/csrc/phylo_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args {
    const double* leaves; double* pool; const int32_t* child; const double* Pmat; const double* pi; double* outll;
    int N, S, K, r;
};

__device__ __forceinline__ const double* child_ptr(const Args& a, int id) {
    const size_t node_sz = (size_t)a.S * 4;
    return id < a.N ? a.leaves + (size_t)id * node_sz : a.pool + (size_t)(id - a.N) * node_sz;
}

// MODE: 0 full, 1 no log, 2 no store, 3 no log + no store, 4 nt store
template <int MODE, int WAVES_PER_EU>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_base(const Args a) {
    __shared__ double sh4[4];
    const int k = blockIdx.x;
    const double* Lp = child_ptr(a, a.child[k * 2]);
    const double* Rp = child_ptr(a, a.child[k * 2 + 1]);
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    double Pl[16], Pr[16];
    const double* P = a.Pmat + (size_t)k * 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) { Pl[j] = P[j]; Pr[j] = P[16 + j]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double col = 0.0;
    for (int s = threadIdx.x; s < a.S; s += 256) {
        double Lv[4], Rv[4], o[4];
        pk_load4(Lp + (size_t)s * 4, Lv);
        pk_load4(Rp + (size_t)s * 4, Rv);
        pk_merge_site(Lv, Rv, Pl, Pr, o);
        if (MODE == 4) pk_store4_nt(out + (size_t)s * 4, o);
        else if (MODE != 2 && MODE != 3) pk_store4(out + (size_t)s * 4, o);
        const double lik = pk_site_lik(pi, o);
        col = col + ((MODE == 1 || MODE == 3) ? lik : pm_log(lik));
    }
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) a.outll[k] = tot;
}

// software-pipelined: the next iteration's children are loaded before the current one is computed
template <int WAVES_PER_EU, bool NT>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_prefetch(const Args a) {
    __shared__ double sh4[4];
    const int k = blockIdx.x;
    const double* Lp = child_ptr(a, a.child[k * 2]);
    const double* Rp = child_ptr(a, a.child[k * 2 + 1]);
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    double Pl[16], Pr[16];
    const double* P = a.Pmat + (size_t)k * 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) { Pl[j] = P[j]; Pr[j] = P[16 + j]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double col = 0.0;
    int s = threadIdx.x;
    double Lv[4], Rv[4], Ln[4], Rn[4], o[4];
    if (s < a.S) { pk_load4(Lp + (size_t)s * 4, Lv); pk_load4(Rp + (size_t)s * 4, Rv); }
    while (s < a.S) {
        const int sn = s + 256;
        if (sn < a.S) { pk_load4(Lp + (size_t)sn * 4, Ln); pk_load4(Rp + (size_t)sn * 4, Rn); }
        pk_merge_site(Lv, Rv, Pl, Pr, o);
        if (NT) pk_store4_nt(out + (size_t)s * 4, o); else pk_store4(out + (size_t)s * 4, o);
        col = col + pm_log(pk_site_lik(pi, o));
#pragma unroll
        for (int j = 0; j < 4; ++j) { Lv[j] = Ln[j]; Rv[j] = Rn[j]; }
        s = sn;
    }
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) a.outll[k] = tot;
}

// all loads of the (<= 4) iterations first, then compute: S <= 1024 only
template <int WAVES_PER_EU, bool NT>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_loadall(const Args a) {
    __shared__ double sh4[4];
    const int k = blockIdx.x;
    const double* Lp = child_ptr(a, a.child[k * 2]);
    const double* Rp = child_ptr(a, a.child[k * 2 + 1]);
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    double Pl[16], Pr[16];
    const double* P = a.Pmat + (size_t)k * 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) { Pl[j] = P[j]; Pr[j] = P[16 + j]; }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double Lv[4][4], Rv[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int s = threadIdx.x + 256 * it;
        if (s < a.S) { pk_load4(Lp + (size_t)s * 4, Lv[it]); pk_load4(Rp + (size_t)s * 4, Rv[it]); }
    }
    double col = 0.0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int s = threadIdx.x + 256 * it;
        if (s < a.S) {
            double o[4];
            pk_merge_site(Lv[it], Rv[it], Pl, Pr, o);
            if (NT) pk_store4_nt(out + (size_t)s * 4, o); else pk_store4(out + (size_t)s * 4, o);
            col = col + pm_log(pk_site_lik(pi, o));
        }
    }
    const double tot = pk_block_canon_sum(col, sh4);
    if (threadIdx.x == 0) a.outll[k] = tot;
}

// ---- half-row form: a lane PAIR owns a site; each lane loads/stores 16 contiguous bytes (states 2h, 2h+1),
// rows are completed through DPP quad_perm moves, each lane produces 2 of the 4 output states.
__device__ __forceinline__ double hb_dpp_even(double v) {   // value held by the even lane of my pair
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xA0, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xA0, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double hb_dpp_odd(double v) {    // value held by the odd lane of my pair
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xF5, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xF5, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int WAVES_PER_EU, bool NT>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_half(const Args a) {
    __shared__ double cols[256];
    __shared__ double sh4[4];
    const int k = blockIdx.x, tid = threadIdx.x, p = tid >> 1, h = tid & 1;
    const double* Lp = child_ptr(a, a.child[k * 2]);
    const double* Rp = child_ptr(a, a.child[k * 2 + 1]);
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    const double* P = a.Pmat + (size_t)k * 32;
    double Plc[4][2], Prc[4][2];                    // my two columns (states 2h, 2h+1) of P_l and P_r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const pk_d2 x = *reinterpret_cast<const pk_d2*>(P + i * 4 + 2 * h);
        const pk_d2 y = *reinterpret_cast<const pk_d2*>(P + 16 + i * 4 + 2 * h);
        Plc[i][0] = x.x; Plc[i][1] = x.y; Prc[i][0] = y.x; Prc[i][1] = y.y;
    }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double col = 0.0;
    const int nq = (a.S + 255) >> 8;
    for (int q = 0; q < nq; ++q) {
        const int sa = p + 256 * q, sb = sa + 128;
        const bool va = sa < a.S, vb = sb < a.S;
        pk_d2 la = {0, 0}, ra = {0, 0}, lb = {0, 0}, rb = {0, 0};
        if (va) { la = *reinterpret_cast<const pk_d2*>(Lp + (size_t)sa * 4 + 2 * h); ra = *reinterpret_cast<const pk_d2*>(Rp + (size_t)sa * 4 + 2 * h); }
        if (vb) { lb = *reinterpret_cast<const pk_d2*>(Lp + (size_t)sb * 4 + 2 * h); rb = *reinterpret_cast<const pk_d2*>(Rp + (size_t)sb * 4 + 2 * h); }
        double lik[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const pk_d2 l2 = t ? lb : la, r2 = t ? rb : ra;
            const double L[4] = {hb_dpp_even(l2.x), hb_dpp_even(l2.y), hb_dpp_odd(l2.x), hb_dpp_odd(l2.y)};
            const double R[4] = {hb_dpp_even(r2.x), hb_dpp_even(r2.y), hb_dpp_odd(r2.x), hb_dpp_odd(r2.y)};
            double o[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double lp = L[0] * Plc[0][c];
                lp = pm_fma(L[1], Plc[1][c], lp);
                lp = pm_fma(L[2], Plc[2][c], lp);
                lp = pm_fma(L[3], Plc[3][c], lp);
                double rp = R[0] * Prc[0][c];
                rp = pm_fma(R[1], Prc[1][c], rp);
                rp = pm_fma(R[2], Prc[2][c], rp);
                rp = pm_fma(R[3], Prc[3][c], rp);
                o[c] = lp * rp;
            }
            const int s = t ? sb : sa;
            if (t ? vb : va) {
                pk_d2 ov = {o[0], o[1]};
                if (NT) __builtin_nontemporal_store(ov, reinterpret_cast<pk_d2*>(out + (size_t)s * 4 + 2 * h));
                else *reinterpret_cast<pk_d2*>(out + (size_t)s * 4 + 2 * h) = ov;
            }
            const double f[4] = {hb_dpp_even(o[0]), hb_dpp_even(o[1]), hb_dpp_odd(o[0]), hb_dpp_odd(o[1])};
            lik[t] = pk_site_lik(pi, f);
        }
        const bool mine = h ? vb : va;                 // even lane: site a -> column p; odd lane: site b -> column p+128
        if (mine) col = col + pm_log(h ? lik[1] : lik[0]);
    }
    cols[p + 128 * h] = col;
    __syncthreads();
    const double tot = pk_block_canon_sum(cols[tid], sh4);
    if (tid == 0) a.outll[k] = tot;
}

template <int WAVES_PER_EU, bool NT>
__global__ __launch_bounds__(256, WAVES_PER_EU) void k_half_pf(const Args a) {
    __shared__ double cols[256];
    __shared__ double sh4[4];
    const int k = blockIdx.x, tid = threadIdx.x, p = tid >> 1, h = tid & 1;
    const double* Lp = child_ptr(a, a.child[k * 2]);
    const double* Rp = child_ptr(a, a.child[k * 2 + 1]);
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    const double* P = a.Pmat + (size_t)k * 32;
    double Plc[4][2], Prc[4][2];                    // my two columns (states 2h, 2h+1) of P_l and P_r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const pk_d2 x = *reinterpret_cast<const pk_d2*>(P + i * 4 + 2 * h);
        const pk_d2 y = *reinterpret_cast<const pk_d2*>(P + 16 + i * 4 + 2 * h);
        Plc[i][0] = x.x; Plc[i][1] = x.y; Prc[i][0] = y.x; Prc[i][1] = y.y;
    }
    const double pi[4] = {a.pi[0], a.pi[1], a.pi[2], a.pi[3]};
    double col = 0.0;
    const int nq = (a.S + 255) >> 8;
    pk_d2 nla = {0, 0}, nra = {0, 0}, nlb = {0, 0}, nrb = {0, 0};
    if (p < a.S) { nla = *reinterpret_cast<const pk_d2*>(Lp + (size_t)p * 4 + 2 * h); nra = *reinterpret_cast<const pk_d2*>(Rp + (size_t)p * 4 + 2 * h); }
    if (p + 128 < a.S) { nlb = *reinterpret_cast<const pk_d2*>(Lp + (size_t)(p + 128) * 4 + 2 * h); nrb = *reinterpret_cast<const pk_d2*>(Rp + (size_t)(p + 128) * 4 + 2 * h); }
    for (int q = 0; q < nq; ++q) {
        const int sa = p + 256 * q, sb = sa + 128;
        const bool va = sa < a.S, vb = sb < a.S;
        const pk_d2 la = nla, ra = nra, lb = nlb, rb = nrb;
        if (sa + 256 < a.S) { nla = *reinterpret_cast<const pk_d2*>(Lp + (size_t)(sa + 256) * 4 + 2 * h); nra = *reinterpret_cast<const pk_d2*>(Rp + (size_t)(sa + 256) * 4 + 2 * h); }
        if (sb + 256 < a.S) { nlb = *reinterpret_cast<const pk_d2*>(Lp + (size_t)(sb + 256) * 4 + 2 * h); nrb = *reinterpret_cast<const pk_d2*>(Rp + (size_t)(sb + 256) * 4 + 2 * h); }
        double lik[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const pk_d2 l2 = t ? lb : la, r2 = t ? rb : ra;
            const double L[4] = {hb_dpp_even(l2.x), hb_dpp_even(l2.y), hb_dpp_odd(l2.x), hb_dpp_odd(l2.y)};
            const double R[4] = {hb_dpp_even(r2.x), hb_dpp_even(r2.y), hb_dpp_odd(r2.x), hb_dpp_odd(r2.y)};
            double o[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double lp = L[0] * Plc[0][c];
                lp = pm_fma(L[1], Plc[1][c], lp);
                lp = pm_fma(L[2], Plc[2][c], lp);
                lp = pm_fma(L[3], Plc[3][c], lp);
                double rp = R[0] * Prc[0][c];
                rp = pm_fma(R[1], Prc[1][c], rp);
                rp = pm_fma(R[2], Prc[2][c], rp);
                rp = pm_fma(R[3], Prc[3][c], rp);
                o[c] = lp * rp;
            }
            const int s = t ? sb : sa;
            if (t ? vb : va) {
                pk_d2 ov = {o[0], o[1]};
                if (NT) __builtin_nontemporal_store(ov, reinterpret_cast<pk_d2*>(out + (size_t)s * 4 + 2 * h));
                else *reinterpret_cast<pk_d2*>(out + (size_t)s * 4 + 2 * h) = ov;
            }
            const double f[4] = {hb_dpp_even(o[0]), hb_dpp_even(o[1]), hb_dpp_odd(o[0]), hb_dpp_odd(o[1])};
            lik[t] = pk_site_lik(pi, f);
        }
        const bool mine = h ? vb : va;                 // even lane: site a -> column p; odd lane: site b -> column p+128
        if (mine) col = col + pm_log(h ? lik[1] : lik[0]);
    }
    cols[p + 128 * h] = col;
    __syncthreads();
    const double tot = pk_block_canon_sum(cols[tid], sh4);
    if (tid == 0) a.outll[k] = tot;
}

// pure store of the output slab (write ceiling at this grid shape)
template <bool NT>
__global__ __launch_bounds__(256) void k_fill(const Args a) {
    const int k = blockIdx.x;
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    double o[4] = {1.0, 2.0, 3.0, (double)k};
    for (int s = threadIdx.x; s < a.S; s += 256) {
        if (NT) pk_store4_nt(out + (size_t)s * 4, o); else pk_store4(out + (size_t)s * 4, o);
    }
}

// store ceiling with 16 contiguous bytes per lane (1 KiB per wave-instruction), block per particle
template <bool NT>
__global__ __launch_bounds__(256) void k_fill16(const Args a) {
    const int k = blockIdx.x;
    double* out = a.pool + ((size_t)a.r * a.K + k) * a.S * 4;
    const int n16 = a.S * 2;                     // 16-byte pieces per node
    pk_d2 v = {1.0, (double)k};
    for (int i = threadIdx.x; i < n16; i += 256) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<pk_d2*>(out) + i);
        else reinterpret_cast<pk_d2*>(out)[i] = v;
    }
}

// same bytes, flat grid-stride over the whole slab (2048 blocks)
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_flat(const Args a) {
    pk_d2* out = reinterpret_cast<pk_d2*>(a.pool + (size_t)a.r * a.K * a.S * 4);
    const size_t n16 = (size_t)a.K * a.S * 2;
    pk_d2 v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

template <typename F>
double time_variant(const char* name, F launch, Args a, int R, const std::vector<int32_t*>& childs, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    for (int w = 0; w < 2; ++w)
        for (int r = 0; r < R; ++r) { a.r = r; a.child = childs[r]; launch(a, st); }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int w = 0; w < reps; ++w)
        for (int r = 0; r < R; ++r) { a.r = r; a.child = childs[r]; launch(a, st); }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * R);
    const double bytes = 96.0 * a.K * a.S;
    printf("%-34s %7.2f us/launch   %7.1f GB/s algorithmic\n", name, us, bytes / us * 1e-3);
    return us;
}

int main(int argc, char** argv) {
    const int N = 12, S = argc > 2 ? atoi(argv[2]) : 898, K = argc > 1 ? atoi(argv[1]) : 2048, R = N - 1;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double *leaves, *pool, *Pmat, *pi, *outll;
    CK(hipMalloc(&leaves, (size_t)N * S * 4 * 8));
    CK(hipMalloc(&pool, (size_t)R * K * S * 4 * 8));
    CK(hipMalloc(&Pmat, (size_t)K * 32 * 8));
    CK(hipMalloc(&pi, 4 * 8));
    CK(hipMalloc(&outll, (size_t)K * 8));
    std::vector<double> h((size_t)N * S * 4);
    srand(1);
    for (auto& v : h) v = 0.05 + (rand() % 1000) / 1000.0;
    CK(hipMemcpy(leaves, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> hp((size_t)K * 32);
    for (auto& v : hp) v = 0.1 + (rand() % 1000) / 2000.0;
    CK(hipMemcpy(Pmat, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
    double hpi[4] = {0.25, 0.25, 0.25, 0.25};
    CK(hipMemcpy(pi, hpi, 32, hipMemcpyHostToDevice));
    // fill the pool with positive numbers
    {
        std::vector<double> slab((size_t)K * S * 4);
        for (auto& v : slab) v = 0.05 + (rand() % 1000) / 1000.0;
        for (int r = 0; r < R; ++r) CK(hipMemcpy(pool + (size_t)r * K * S * 4, slab.data(), slab.size() * 8, hipMemcpyHostToDevice));
    }
    // child tables.  mode A ("sweep-like"): at rank r a child is a leaf w.p. (N-2r)/(N-r) else one of a FEW
    // distinct ancestors' nodes of earlier ranks (degenerate resampling).  mode B ("spread"): internal
    // children uniformly spread over all earlier nodes (worst case for caches).
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int32_t*> childs(R);
        for (int r = 0; r < R; ++r) {
            std::vector<int32_t> c((size_t)K * 2);
            for (int k = 0; k < K; ++k)
                for (int j = 0; j < 2; ++j) {
                    const bool leaf = (r == 0) || (rand() % (N - r)) < (N - 2 * r > 0 ? N - 2 * r : 1);
                    int id;
                    if (leaf) id = rand() % N;
                    else {
                        const int rho = rand() % r;
                        const int kap = mode == 0 ? (rand() % 8) * 97 % K : rand() % K;
                        id = N + rho * K + kap;
                    }
                    c[(size_t)k * 2 + j] = id;
                }
            CK(hipMalloc(&childs[r], c.size() * 4));
            CK(hipMemcpy(childs[r], c.data(), c.size() * 4, hipMemcpyHostToDevice));
        }
        Args a{leaves, pool, nullptr, Pmat, pi, outll, N, S, K, 0};
        printf("== children: %s  (K=%d S=%d) ==\n", mode == 0 ? "sweep-like (few live ancestors)" : "spread over all earlier nodes", K, S);
#define RUN(name, kern) time_variant(name, [&](Args x, hipStream_t s) { hipLaunchKernelGGL((kern), dim3(K), dim3(256), 0, s, x); }, a, R, childs, st)
        RUN("base w6", (k_base<0, 6>));
        RUN("base w8", (k_base<0, 8>));
        RUN("base w4", (k_base<0, 4>));
        RUN("base nt-store w6", (k_base<4, 6>));
        RUN("base nt-store w8", (k_base<4, 8>));
        RUN("no log w6", (k_base<1, 6>));
        RUN("no store w6", (k_base<2, 6>));
        RUN("no log no store w6", (k_base<3, 6>));
        RUN("prefetch w6", (k_prefetch<6, false>));
        RUN("prefetch nt w6", (k_prefetch<6, true>));
        RUN("prefetch nt w8", (k_prefetch<8, true>));
        RUN("prefetch nt w4", (k_prefetch<4, true>));
        if (S <= 1024) {
            RUN("loadall nt w4", (k_loadall<4, true>));
            RUN("loadall nt w6", (k_loadall<6, true>));
            RUN("loadall w4", (k_loadall<4, false>));
        }
        RUN("half-row w6", (k_half<6, false>));
        RUN("half-row nt w6", (k_half<6, true>));
        RUN("half-row pf nt w6", (k_half_pf<6, true>));
        RUN("half-row pf nt w4", (k_half_pf<4, true>));
        RUN("half-row pf w4", (k_half_pf<4, false>));
        RUN("half-row nt w4", (k_half<4, true>));
        {   // bit-for-bit check of the half-row form against the base kernel on rank R-1
            std::vector<double> o1((size_t)K), o2((size_t)K), n1((size_t)K * S * 4), n2((size_t)K * S * 4);
            Args b = a; b.r = R - 1; b.child = childs[R - 1];
            hipLaunchKernelGGL((k_base<0, 6>), dim3(K), dim3(256), 0, st, b);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(o1.data(), outll, K * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(n1.data(), pool + (size_t)(R - 1) * K * S * 4, n1.size() * 8, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL((k_half<6, true>), dim3(K), dim3(256), 0, st, b);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(o2.data(), outll, K * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(n2.data(), pool + (size_t)(R - 1) * K * S * 4, n2.size() * 8, hipMemcpyDeviceToHost));
            size_t bad = 0, badn = 0;
            for (int i = 0; i < K; ++i) bad += memcmp(&o1[i], &o2[i], 8) != 0;
            for (size_t i = 0; i < n1.size(); ++i) badn += memcmp(&n1[i], &n2[i], 8) != 0;
            printf("half-row vs base: %zu of %d sums differ, %zu of %zu partials differ\n", bad, K, badn, n1.size());
        }
        RUN("fill (store only)", (k_fill<false>));
        RUN("fill nt (store only)", (k_fill<true>));
        RUN("fill16 (16B/lane coalesced)", (k_fill16<false>));
        RUN("fill16 nt", (k_fill16<true>));
        RUN("fill flat grid-stride", (k_fill_flat<false>));
        RUN("fill flat nt", (k_fill_flat<true>));
        for (int r = 0; r < R; ++r) CK(hipFree(childs[r]));
    }
    return 0;
}
