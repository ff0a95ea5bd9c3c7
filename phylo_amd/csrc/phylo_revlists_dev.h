// phylo_revlists_dev.h -- the integer lists of the reverse pass built ON THE DEVICE (round 3; phylo_revlists.h is the host form, kept
// for the twisted proposal, the tile form, sweeps without marks, K > 8192, and as the test reference).  Plain proposal, rows form,
// after a lazy sweep (the early pg_nodes_free): who adopted whom per rank event, which nodes have which parents, heavy nodes'
// chunks, the flagged nodes by rank event -- in the layout pg_lists_carve gives the slab, so the reverse kernels do not care who
// built it.
//
//   pg_dl_adopters one workgroup per rank event: a STABLE block radix sort (rocPRIM's block primitive) of the K adopters by ancestor
//                  is ad_idx's row (ascending adopter within an ancestor, like the host's counting sort); the sorted keys, searched
//                  per ancestor, give ad_off's row and who was adopted at all.  No atomics: a few ancestors take nearly every draw,
//                  and K increments of one counter are K serial round trips;
//   pg_dl_count    parents per node: integer atomics (the counts are exact whatever the order), one per distinct child of a wave
//                  (lanes with the same child add up first: the children repeat like the ancestors);
//   pg_dl_sums     per block of 1024 nodes the four sums that become offsets: adopted nodes, parents, chunks, flagged nodes;
//   pg_dl_lists    thread = node: its four exclusive prefixes (the sums of the blocks before its own, a scan within the block) and
//                  everything that follows from them -- adp, par_off, heavy, chunk_beg / chunk_cnt, slow_flag, slow_idx, the starts
//                  of every rank event (meta) -- and the sort input of the node's two parent entries; the block that finishes last
//                  copies meta to pinned host memory in one go: ~50 integers are all the host waits for;
//   a STABLE device sort (rocPRIM, launched by the host code, captured in a hipGraph: six small launches): (child, parent is
//                  flagged) -> entry, whose output IS par_idx: equal keys keep their input order, ascending entry, so the order of
//                  every gather is fixed and gradients are bit-reproducible run to run.
// Two differences from the host form, both invisible to the kernels: heavy[] holds the chunk's GLOBAL index (the host form: within
// the rank event; the caller passes chunk0 = 0), and a node's flagged parents end its list in ASCENDING order (host: descending,
// its cursor runs from the back): the same sums in another fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/block/block_radix_sort.hpp>

#include "phylo_revlists.h"

struct pg_dl_args {
    int N, R, K;
    const int64_t* anc;                // [R-1][K] ancestors (device)
    const int32_t* child;              // [R][K][2] node ids (device)
    int32_t *cnt_par, *ticket;         // [R K], [1]: zeroed by pg_dl_adopters
    int32_t* adopted;                  // [R K]: somebody adopted the node (pg_dl_adopters writes every row)
    int32_t* bsum;                     // [blocks][4]
    uint32_t *pkey, *pval;             // [2 R K]: sort input of the parents' entries
    pg_lists L;                        // the slab (device pointers)
    int32_t *dmeta, *meta;             // device, pinned host: ev_adp0[R + 1] | ev_slow0[R + 1] | n_adp, n_chunks, n_slow, n_par
};
#define PG_DL_META_INTS(R) (2 * ((R) + 1) + 4)
#define PG_DL_BLOCK 1024
#define PG_DL_MAX_K 8192               // pg_dl_adopters: 1024 threads x 8 adopters

template <int ITEMS>
struct pg_dl_sort {
    typedef rocprim::block_radix_sort<unsigned int, PG_DL_BLOCK, ITEMS, unsigned int> type;
    static constexpr size_t storage_bytes = (sizeof(typename type::storage_type) + 15) & ~(size_t)15;
};

// grid R (workgroup = rank event), dynamic LDS: the sort's storage, then K sorted keys.  bits = bit length of K (the padding key).
template <int ITEMS>
__global__ __launch_bounds__(PG_DL_BLOCK) void pg_dl_adopters(const pg_dl_args a, unsigned bits) {
    extern __shared__ __align__(16) unsigned char pg_dl_lds[];
    typedef typename pg_dl_sort<ITEMS>::type sort_t;
    typename sort_t::storage_type& st = *reinterpret_cast<typename sort_t::storage_type*>(pg_dl_lds);
    unsigned int* skeys = reinterpret_cast<unsigned int*>(pg_dl_lds + pg_dl_sort<ITEMS>::storage_bytes);
    const int r = blockIdx.x, K = a.K, tid = threadIdx.x;
    for (int x = tid; x < K; x += PG_DL_BLOCK) a.cnt_par[(size_t)r * K + x] = 0;   // what pg_dl_count and pg_dl_lists count into
    if (r == 0) {                                            // nobody adopts at rank event 0, nobody adopts the last rank event's nodes
        if (tid == 0) *a.ticket = 0;
        for (int x = tid; x <= K; x += PG_DL_BLOCK) a.L.ad_off[x] = 0;
        for (int x = tid; x < K; x += PG_DL_BLOCK) a.adopted[(size_t)(a.R - 1) * K + x] = 0;
        return;
    }
    const int64_t* anc = a.anc + (size_t)(r - 1) * K;
    unsigned int key[ITEMS], val[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int k = tid * ITEMS + j;
        key[j] = k < K ? (unsigned int)anc[k] : (unsigned int)K;
        val[j] = (unsigned int)k;
    }
    sort_t().sort(key, val, st, 0u, bits);                  // stable: ascending adopter within an ancestor
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int p = tid * ITEMS + j;
        if (p < K) {
            skeys[p] = key[j];
            a.L.ad_idx[(size_t)r * K + p] = (int32_t)val[j];
        }
    }
    __syncthreads();
    for (int x = tid; x <= K; x += PG_DL_BLOCK) {            // first position with key >= x
        int lo = 0, hi = K;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skeys[mid] < (unsigned int)x) lo = mid + 1; else hi = mid;
        }
        a.L.ad_off[(size_t)r * (K + 1) + x] = lo;
        if (x < K) a.adopted[(size_t)(r - 1) * K + x] = lo < K && skeys[lo] == (unsigned int)x;
    }
}

__global__ __launch_bounds__(256) void pg_dl_count(const pg_dl_args a) {
    const long nn = (long)a.R * a.K, e = 2L * a.K + (long)blockIdx.x * blockDim.x + threadIdx.x;   // (rank event 0 merges leaves)
    const int lane = threadIdx.x & 63;
    int ch = -1;
    if (e < 2 * nn) {
        ch = a.child[e];
        if (ch < a.N) ch = -1;
    }
    unsigned long long todo = __ballot(ch >= 0);
    while (todo) {                                           // one add per distinct child of the wave
        const int lead = __ffsll((long long)todo) - 1;
        const int lch = __shfl(ch, lead, 64);
        const unsigned long long same = __ballot(ch == lch);
        if (lane == lead) atomicAdd(&a.cnt_par[lch - a.N], (int)__popcll(same));
        todo &= ~same;
    }
}

struct pg_dl4 { int v[4]; };
__device__ __forceinline__ pg_dl4 pg_dl_node_counts(const pg_dl_args& a, long i, long nn) {
    pg_dl4 o{};
    if (i < nn) {
        const int np = a.cnt_par[i], ad = a.adopted[i] != 0;
        o.v[0] = ad;
        o.v[1] = np;
        o.v[2] = np > PG_PCHUNK ? (np + PG_HCHUNK - 1) / PG_HCHUNK : 0;
        o.v[3] = (np > 0) | ad;
    }
    return o;
}

__global__ __launch_bounds__(PG_DL_BLOCK) void pg_dl_sums(const pg_dl_args a) {
    __shared__ int sh[4];
    const long nn = (long)a.R * a.K, i = (long)blockIdx.x * PG_DL_BLOCK + threadIdx.x;
    if (threadIdx.x < 4) sh[threadIdx.x] = 0;
    __syncthreads();
    const pg_dl4 c = pg_dl_node_counts(a, i, nn);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int v = c.v[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[q], v);
    }
    __syncthreads();
    if (threadIdx.x < 4) a.bsum[blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x];
}

__global__ __launch_bounds__(PG_DL_BLOCK) void pg_dl_lists(const pg_dl_args a) {
    __shared__ int base[4];
    __shared__ int wsum[4][PG_DL_BLOCK / 64];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, K = a.K, R = a.R;
    const long nn = (long)R * K, i = (long)blockIdx.x * PG_DL_BLOCK + tid;
    if (tid < 4) base[tid] = 0;
    __syncthreads();
    {   // the sums of the blocks before this one (a few hundred at most: a strided add per thread, one LDS atomic per wave)
        int part[4] = {0, 0, 0, 0};
        for (int b = tid; b < (int)blockIdx.x; b += PG_DL_BLOCK)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[q] += a.bsum[b * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int v = part[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&base[q], v);
        }
    }
    const pg_dl4 c = pg_dl_node_counts(a, i, nn);
    int inc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                            // inclusive scan within the wave, the waves' totals through LDS
        int v = c.v[q];
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(v, off, 64);
            if (lane >= off) v += t;
        }
        inc[q] = v;
        if (lane == 63) wsum[q][wv] = v;
    }
    __syncthreads();
    int ex[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int before = base[q];
        for (int w = 0; w < wv; ++w) before += wsum[q][w];
        ex[q] = before + inc[q] - c.v[q];
    }
    int32_t* ev_adp0 = a.dmeta;
    int32_t* ev_slow0 = a.dmeta + (R + 1);
    if (i < nn) {
        const int r = (int)(i / K), x = (int)(i - (long)r * K);
        const int ad = c.v[0], np = c.v[1], nch = c.v[2], f0 = (np > 0 ? 1 : 0) | (ad ? 4 : 0);
        if (x == 0) {                                        // a rank event starts here
            ev_adp0[r] = ex[0];
            ev_slow0[r] = ex[3];
        }
        if (ad) a.L.adp[ex[0]] = (int32_t)i;
        a.L.par_off[i] = ex[1];
        a.L.heavy[i] = nch ? ex[2] : -1;
        for (int q = 0, b = ex[1]; q < nch; ++q, b += PG_HCHUNK) {
            a.L.chunk_beg[ex[2] + q] = b;
            a.L.chunk_cnt[ex[2] + q] = ex[1] + np - b < PG_HCHUNK ? ex[1] + np - b : PG_HCHUNK;
        }
        a.L.slow_flag[i] = f0 ? (f0 | (ex[3] << 3)) : 0;
        if (f0) a.L.slow_idx[ex[3]] = (int32_t)i;
        // the node's two entries as somebody's parent: key (child, this node is flagged), leaves behind everything
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const long e = 2 * i + side;
            const int ch = a.child[e];
            const bool in = r >= 1 && ch >= a.N;
            a.pkey[e] = in ? (((uint32_t)(ch - a.N) << 1) | (f0 ? 1u : 0u)) : (uint32_t)(2 * nn);
            a.pval[e] = (uint32_t)e | (f0 ? 0u : (uint32_t)PG_FREE_PARENT);
        }
        if (i == nn - 1) {                                   // totals
            const int n_adp = ex[0] + ad, n_chunks = ex[2] + nch, n_slow = ex[3] + c.v[3], n_par = ex[1] + np;
            ev_adp0[R] = n_adp;
            ev_slow0[R] = n_slow;
            a.L.par_off[nn] = n_par;
            int32_t* t = a.dmeta + 2 * (R + 1);
            t[0] = n_adp; t[1] = n_chunks; t[2] = n_slow; t[3] = n_par;
        }
    }
    // the block that finishes last hands meta to the host: one run of stores to pinned memory
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(a.ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (last) {
        __threadfence();
        for (int q = tid; q < PG_DL_META_INTS(R); q += PG_DL_BLOCK) a.meta[q] = __hip_atomic_load(a.dmeta + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
