// phylo_revlists.h -- the integer lists of the reverse pass (phylo_sweep_backward), built on the HOST from the ancestors and the
// children that the sweep left: who adopted whom (per rank event, counting sort by ancestor) and which nodes have which parents
// (entries node * 2 + side grouped by child), plus what follows from them per node: heavy nodes cut into chunks, the nodes that go
// through pg_nodes_rows (flags, lists by rank event), the adopted particles of every rank event.  Plain C++, no HIP: the same
// functions run under phylo_debug_reverse_lists for the CPU tests (tests/test_revlists_cpu.py checks them against a restatement in
// NumPy).  The layout of the slab is the one the device reads (pg_args in phylo_grad.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#define PG_PCHUNK 8                    // parents staged in LDS at a time; more parents than this = a heavy node
#define PG_HCHUNK 32                   // parents per chunk of a heavy node
// par_idx entry: (parent node * 2 + side) | PG_FREE_PARENT when the parent's adjoint row is the own term alone and is not stored
// (rows form: pg_nodes_free); the gather then recomputes it from alpha_parent
#define PG_FREE_PARENT (1 << 30)

struct pg_lists {
    int32_t *ad_off, *ad_idx;          // [R][K+1], [R][K]
    int32_t *par_off, *par_idx;        // [R K + 1], [2 R K]
    int32_t *heavy, *chunk_beg, *chunk_cnt;   // [R K], [cap], [cap]
    int32_t *slow_flag, *slow_idx;     // [R K] x 2
    int32_t* adp;                      // [R K]
    size_t cap;
};
inline size_t pg_lists_cap(size_t R, size_t K) { return 2 * R * K / 4 + 1; }
inline size_t pg_lists_ints(size_t R, size_t K) {
    return R * (K + 1) + R * K + (R * K + 1) + 2 * R * K + R * K + 2 * pg_lists_cap(R, K) + 3 * R * K;
}
inline pg_lists pg_lists_carve(int32_t* base, size_t R, size_t K) {
    const size_t nn = R * K;
    pg_lists L;
    L.cap = pg_lists_cap(R, K);
    L.ad_off = base;
    L.ad_idx = L.ad_off + R * (K + 1);
    L.par_off = L.ad_idx + nn;
    L.par_idx = L.par_off + nn + 1;
    L.heavy = L.par_idx + 2 * nn;
    L.chunk_beg = L.heavy + nn;
    L.chunk_cnt = L.chunk_beg + L.cap;
    L.slow_flag = L.chunk_cnt + L.cap;
    L.slow_idx = L.slow_flag + nn;
    L.adp = L.slow_idx + nn;
    return L;
}

// Clears what the builders count into (ad_off, ad_idx, par_off, slow_flag).  Between this and pg_build_parents the caller may set
// bit 1 of slow_flag[x] (twisted proposal: node x has look-ahead entries).
inline void pg_lists_clear(const pg_lists& L, int R, int K) {
    const size_t nn = (size_t)R * K;
    memset(L.ad_off, 0, ((size_t)R * (K + 1) + nn + nn + 1) * 4);      // ad_off, ad_idx, par_off
    memset(L.slow_flag, 0, nn * 4);
}

// Adopters of every particle at every rank event (anc[r-1][k'] = the particle k' adopted at rank event r; ascending k' within a
// list), and the adopted particles (r * K + k), grouped by rank event: adp[ev_adp0[r] .. ev_adp0[r+1]).  Returns their number.
// A few ancestors take nearly all the draws, so counters and cursors are chains of store-to-load forwards on one address: the
// particles are taken as four contiguous quarters with a counter row each (four independent chains), whose prefix sums give
// every quarter its own cursor into an ancestor's list.
inline int32_t pg_build_adopters(int R, int K, const int64_t* anc, const pg_lists& L, std::vector<int32_t>& cur, std::vector<int32_t>& ev_adp0) {
    ev_adp0.assign((size_t)R + 1, 0);
    int32_t n_adp = 0;
    const int Kq = K / 4;
    cur.assign((size_t)4 * K, 0);
    for (int r = 1; r < R; ++r) {
        int32_t* off = L.ad_off + (size_t)r * (K + 1);
        const int64_t* a = anc + (size_t)(r - 1) * K;
        int32_t* idx = L.ad_idx + (size_t)r * K;
        int32_t *c0 = cur.data(), *c1 = c0 + K, *c2 = c1 + K, *c3 = c2 + K;
        if (r > 1) memset(c0, 0, (size_t)4 * K * 4);
        for (int k = 0; k < Kq; ++k) {
            ++c0[a[k]]; ++c1[a[k + Kq]]; ++c2[a[k + 2 * Kq]]; ++c3[a[k + 3 * Kq]];
        }
        for (int k = 4 * Kq; k < K; ++k) ++c3[a[k]];       // (the last quarter takes the remainder)
        ev_adp0[r - 1] = n_adp;
        int32_t run = 0;
        for (int x = 0; x < K; ++x) {
            const int32_t t0 = c0[x], t1 = c1[x], t2 = c2[x], t3 = c3[x];
            off[x] = run;
            c0[x] = run; c1[x] = run + t0; c2[x] = run + t0 + t1; c3[x] = run + t0 + t1 + t2;
            const int32_t tot = (t0 + t1) + (t2 + t3);
            if (tot) L.adp[n_adp++] = (r - 1) * K + x;      // somebody adopts (r - 1, x) at rank event r
            run += tot;
        }
        off[K] = run;
        for (int k = 0; k < Kq; ++k) {
            idx[c0[a[k]]++] = k; idx[c1[a[k + Kq]]++] = k + Kq; idx[c2[a[k + 2 * Kq]]++] = k + 2 * Kq; idx[c3[a[k + 3 * Kq]]++] = k + 3 * Kq;
        }
        for (int k = 4 * Kq; k < K; ++k) idx[c3[a[k]]++] = k;
    }
    if (R >= 1) ev_adp0[R - 1] = n_adp;
    ev_adp0[R] = n_adp;
    if (R == 1) ev_adp0[0] = 0;
    return n_adp;
}

// Bit 2 of slow_flag for every node somebody adopted (r - 1, anc[r-1][k]): what pg_build_parents needs of the adopters when the early
// pg_nodes_free has skipped the adopted nodes -- one pass over the ancestors, so the parents' lists can be built (and the launch that
// needs them started) before the adopters' counting sorts.
inline void pg_mark_adopted(int R, int K, const int64_t* anc, const pg_lists& L) {
    for (int r = 1; r < R; ++r) {
        const int64_t* a = anc + (size_t)(r - 1) * K;
        int32_t* f = L.slow_flag + (size_t)(r - 1) * K;
        for (int k = 0; k < K; ++k) f[a[k]] |= 4;
    }
}

struct pg_parents_info {
    size_t n_chunks, max_chunks;       // chunks in all, most chunks of one rank event
    int32_t n_slow, n_par;             // flagged nodes, parent entries
};

// (Tried on the builders, measured on a GPU box, not kept: four counter rows per child as in pg_build_adopters -- faster on synthetic
//  genealogies with many internal children, 0.18 -> 0.26 ms on primate.p's, whose later rank events still merge mostly leaves, because
//  of the four times larger cursor array; worker threads -- rank events in groups, quarters of the child entries on four cores --
//  0.6 -> 1.7 ms for DS1 at K = 4096 on the 16-core share of a box.)
// Parents: entries e = node * 2 + side grouped by child (ascending e).  One pass over the nodes turns the counts into offsets and
// decides everything per node: heavy nodes (more than PG_PCHUNK parents) get their list cut into chunks of PG_HCHUNK, numbered
// within the rank event (rank_chunk0); nodes with parents (bit 0), look-ahead entries (bit 1, set by the caller) or -- after the
// early pg_nodes_free -- adopters (bit 2) are flagged ((index in slow_idx) << 3 | bits) and listed by rank event (ev_slow0) for
// pg_nodes_rows; all the others: pg_nodes_free.  A parent that goes through pg_nodes_free never stores its adjoint row: its
// entries carry PG_FREE_PARENT (rows form only).  After the early pg_nodes_free the caller runs pg_mark_adopted first.
inline pg_parents_info pg_build_parents(int N, int R, int K, const int32_t* child, bool rows_form, bool tail_flagged, const pg_lists& L,
                                        std::vector<int32_t>& cur, std::vector<int32_t>& rank_chunk0, std::vector<int32_t>& ev_slow0) {
    const size_t nn = (size_t)R * K;
    // (leaf or internal child is a coin toss in the later rank events: no branch on it -- a leaf counts into one of 64 dummies in
    //  turn: increments of one address are a chain of store-to-load forwards, 5 cycles each)
    int32_t dummy[64] = {0};
    const size_t e0 = 2 * (size_t)K;                        // (the children of rank event 0 are leaves: nothing to count)
    for (size_t e = e0; e < 2 * nn; ++e) {
        const int32_t ch = child[e];
        int32_t* p = ch >= N ? L.par_off + (size_t)(ch - N) + 1 : dummy + (e & 63);
        ++*p;
    }
    rank_chunk0.assign((size_t)R + 1, 0);
    ev_slow0.assign((size_t)R + 1, 0);
    size_t max_chunks = 0, n_chunks = 0;
    int32_t ns = 0, run = 0;
    for (int r = 0; r < R; ++r) {
        rank_chunk0[r] = (int32_t)n_chunks;
        ev_slow0[r] = ns;
        for (int k = 0; k < K; ++k) {
            const size_t x = (size_t)r * K + k;
            const int32_t np = L.par_off[x + 1];         // still the count: the offsets are written behind the read position
            L.par_off[x] = run;
            int32_t f = L.slow_flag[x];                  // (set by the caller: bit 1 look-ahead entries, bit 2 adopted -- pg_mark_adopted:
            if (np) f |= 1;                              //  the early launch skipped those; without parents they join the flagged ones)
            L.heavy[x] = -1;
            if (np > PG_PCHUNK) {
                L.heavy[x] = (int32_t)(n_chunks - rank_chunk0[r]);
                for (int32_t b = run; b < run + np; b += PG_HCHUNK) {
                    L.chunk_beg[n_chunks] = b;
                    L.chunk_cnt[n_chunks] = run + np - b < PG_HCHUNK ? run + np - b : PG_HCHUNK;
                    ++n_chunks;
                }
            }
            if (f) {
                f |= ns << 3;
                L.slow_idx[ns++] = (int32_t)x;
            }
            L.slow_flag[x] = f;
            run += np;
        }
        if (n_chunks - rank_chunk0[r] > max_chunks) max_chunks = n_chunks - rank_chunk0[r];
    }
    L.par_off[nn] = run;
    rank_chunk0[R] = (int32_t)n_chunks;
    ev_slow0[R] = ns;
    // scatter without a branch on leaf / internal (masks, not ?: -- the compiler made a branch of that, mispredicted every other
    // time in the later rank events): a leaf child advances one of 64 dummy cursors and writes into the tail of par_idx,
    // which is never used (the 2 K children of rank event 0 are all leaves).
    // tail_flagged (the plain reverse pass after the early pg_nodes_free): the entries of FREE parents (no flag: their adjoint is
    // recomputed from omega, nothing of the chain is needed) fill a child's
    // list from the front, ascending; the few entries of flagged parents from the back, so the list ends with them in descending
    // order -- pg_parent_chunks_all sums the free ones of every rank event in one launch, pg_nodes_rows walks the tail.
    cur.resize((tail_flagged ? 2 : 1) * (nn + 64));
    int32_t* front = cur.data();
    memcpy(front, L.par_off, nn * 4);
    memset(front + nn, 0, 64 * 4);                          // the 64 dummy cursors start from 0 at every call (their values are masked
    if (tail_flagged) {                                     //  out of the result; left alone they would count up from call to call)
        int32_t* back = front + nn + 64;                    // (the cursors from the back: the row behind the front cursors)
        for (size_t x = 0; x < nn; ++x) back[x] = L.par_off[x + 1] - 1;
        memset(back + nn, 0, 64 * 4);
    }
    const int32_t free_bit = rows_form ? PG_FREE_PARENT : 0;
    const int32_t tail = (int32_t)(2 * nn) - 1;
    int32_t tmask = 1;                                      // dummy slots: the last min(64, 2 K rounded down to a power of two)
    while (tmask * 2 <= 2 * K && tmask < 64) tmask *= 2;
    tmask -= 1;
    const ptrdiff_t rowlen = (ptrdiff_t)(nn + 64);
    for (size_t e = e0; e < 2 * nn; ++e) {                 // e = node * 2 + side, ascending
        const int32_t ch = child[e];
        const int32_t in = -(int32_t)(ch >= N);              // all ones: internal child
        const int32_t lane = (int32_t)(e & 63);
        const int32_t ci = ((ch - N) & in) | (((int32_t)nn + lane) & ~in);
        const int32_t fl = L.slow_flag[e >> 1] != 0;         // the parent is a flagged node
        const int32_t tob = fl & (tail_flagged ? 1 : 0);     // 1: from the back
        int32_t* cp = front + (ptrdiff_t)tob * rowlen + ci;
        const int32_t pos = *cp;
        *cp = pos + 1 - 2 * tob;
        const int32_t di = (pos & in) | ((tail - (lane & tmask)) & ~in);
        L.par_idx[di] = (int32_t)e | (fl ? 0 : free_bit);
    }
    pg_parents_info o;
    o.n_chunks = n_chunks; o.max_chunks = max_chunks; o.n_slow = ns; o.n_par = run;
    return o;
}
