// phylo_hip.hip -- C ABI (include/phylo_hip.h) over the gfx950 kernels in phylo_kernels.h.
// One context = one GPU, one stream.  No CPU fallback: without a HIP device every entry point fails.
#include "../../include/phylo_hip.h"

#include <cstring>                     // (ahead of the HIP headers: rocPRIM's use memcpy unqualified)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>   // the two sorts of phylo_revlists_dev.h

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "phylo_comm.h"
#include "phylo_kernels.h"
#include "phylo_persist.h"
#include "phylo_grad.h"
#include "phylo_revlists_dev.h"
#include "phylo_train.h"

namespace {

thread_local std::string g_last_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct sweep_run {                       // a sweep being issued rank event by rank event
    uint64_t seed = 0;
    uint32_t flags = 0;
    int M = 1, launches = 0, next_r = 0;
    bool twist = false, graph = false, lazy = false, timek = false, fuse_scan = false, active = false, local_book = false;
    bool mat_by_draws = false;             // sharded: owners find their adopted nodes from the draws (no pk_all_marks)
    bool logz_done = false;                // the last scan summed the log-normalisers (no pk_logz_total launch)
    bool book_mat = false;                 // bookkeeping and the writes of the adopted nodes share one launch (pk_rank_book_mat)
    int G = 1;                             // independent sweeps batched in this context (phylo_sweep_batch_async)
    bool final_missing = false;            // the last rank event's nodes were not stored
    int a_done_r = -1;                     // rank event whose first half (sweep_step_a) has been issued
};

// A/B switches of DESIGN.md section 6b, read from the environment ONCE (phylo_create): none changes a result bit
struct env_switches {
    bool eager_nodes = false, rehearse_sharded = false, replicated_book = false, fuse_scan = false,
         book_one_per_wave = false, merge_pair_form = false, no_leaf_codes = false, one_launch = false,
         persist_stamps = false, separate_materialise = false, grad_one_stream = false, grad_two_streams = false, rev_host_lists = false, no_remote_cache = false, no_spin_wait = false,
         no_p2p = false, book_lp16 = false, no_sorted_draws = false, grad_quad_chunks = false, grad_rows_chain = false, grad_rows_no_overlap = false, grad_coeff_chain = false, grad_sort_late = false;
    int persist_wgs = 0;                 // PHYLO_PERSIST_WGS: resident workgroups of the one-launch sweep (0 = default)
    unsigned long long p2p_wait_ticks = PK_P2P_WAIT_TICKS;   // PHYLO_P2P_WAIT_S: bound of a flag wait of the device-side exchange
    size_t p2p_copy_words = 65536;       // PHYLO_P2P_COPY_WORDS: exchanges beyond this many doubles copy with many workgroups (tests lower it)
    int persist_nt = 256;                // PHYLO_PERSIST_NT: threads per workgroup of the one-launch sweep (256 or 512)
    int remote_cache_cap = 0;            // PHYLO_REMOTE_CACHE_CAP: slots of the local cache of remote nodes (0 = 512 MB worth; tests lower it)
    int scan_multi_min = 4096;           // PHYLO_SCAN_MULTI_MIN: groups of more weights than this are scanned by several workgroups
    void read() {
        eager_nodes = getenv("PHYLO_EAGER_NODES") != nullptr;
        rehearse_sharded = getenv("PHYLO_REHEARSE_SHARDED") != nullptr;
        replicated_book = getenv("PHYLO_REPLICATED_BOOK") != nullptr;
        fuse_scan = getenv("PHYLO_FUSE_SCAN") != nullptr;
        book_one_per_wave = getenv("PHYLO_BOOK_ONE_PER_WAVE") != nullptr;
        book_lp16 = getenv("PHYLO_BOOK_LP16") != nullptr;
        no_sorted_draws = getenv("PHYLO_NO_SORTED_DRAWS") != nullptr;
        grad_quad_chunks = getenv("PHYLO_GRAD_QUAD_CHUNKS") != nullptr;
        grad_rows_chain = getenv("PHYLO_GRAD_ROWS_CHAIN") != nullptr;
        grad_rows_no_overlap = getenv("PHYLO_GRAD_ROWS_NO_OVERLAP") != nullptr;
        grad_coeff_chain = getenv("PHYLO_GRAD_COEFF_CHAIN") != nullptr;
        grad_sort_late = getenv("PHYLO_GRAD_SORT_LATE") != nullptr;
        merge_pair_form = getenv("PHYLO_MERGE_PAIR_FORM") != nullptr;
        no_leaf_codes = getenv("PHYLO_NO_LEAF_CODES") != nullptr;
        one_launch = getenv("PHYLO_ONE_LAUNCH") != nullptr;
        persist_stamps = getenv("PHYLO_PERSIST_STAMPS") != nullptr;
        separate_materialise = getenv("PHYLO_SEPARATE_MATERIALISE") != nullptr;
        grad_one_stream = getenv("PHYLO_GRAD_ONE_STREAM") != nullptr;
        grad_two_streams = getenv("PHYLO_GRAD_TWO_STREAMS") != nullptr;
        rev_host_lists = getenv("PHYLO_REV_HOST_LISTS") != nullptr;
        no_remote_cache = getenv("PHYLO_NO_REMOTE_CACHE") != nullptr;
        { const char* e = getenv("PHYLO_REMOTE_CACHE_CAP"); remote_cache_cap = e ? atoi(e) : 0; }
        { const char* e = getenv("PHYLO_SCAN_MULTI_MIN"); scan_multi_min = e ? atoi(e) : 4096; }
        no_spin_wait = getenv("PHYLO_NO_SPIN_WAIT") != nullptr;
        { const char* e = getenv("PHYLO_P2P"); no_p2p = e && atoi(e) == 0; }
        { const char* e = getenv("PHYLO_P2P_COPY_WORDS"); p2p_copy_words = e ? (size_t)atol(e) : 65536; }
        { const char* e = getenv("PHYLO_P2P_WAIT_S"); p2p_wait_ticks = e && atof(e) > 0 ? (unsigned long long)(atof(e) * 1e8) : PK_P2P_WAIT_TICKS; }
        const char* w = getenv("PHYLO_PERSIST_WGS");
        persist_wgs = w ? atoi(w) : 0;
        const char* t = getenv("PHYLO_PERSIST_NT");
        persist_nt = (t && atoi(t) == 512) ? 512 : 256;
    }
};

struct phylo_ctx {
    int device = 0;
    env_switches env;
    int n_cus = 0;                       // compute units of the device
    // one-launch sweep (phylo_persist.h)
    unsigned long long* d_rdraw = nullptr;   // [(N-1)][K] resampling draws
    unsigned long long* d_pctr = nullptr;    // [PK_MAX_GROUPS][PP_CTR_STRIDE] monotone arrival counters
    unsigned long long pctr_base = 0;        // their common value (every group receives Wg arrivals per rank event)
    int pctr_Wg = 0, pctr_G = 0;             // workgroups per group / groups the counters were last used with
    bool pctr_dirty = false;                 // a bounded wait timed out: the counters hold partial arrivals
    int persist_blocks_per_cu = -1;          // occupancy of pp_sweep (-1: not asked yet)
    bool last_persistent = false;
    unsigned long long* d_stamps = nullptr;  // phase stamps of the one-launch sweep (PHYLO_PERSIST_STAMPS=1)
    int K = 0, N = 0, S = 0, A = 4;      // K = global particle count
    int Kloc = 0, k0 = 0;                // this rank's shard
    int rank = 0, world = 1;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;         // per-merge-launch events (PHYLO_TIME_KERNELS)
    std::string err;
    bool have_leaves = false, have_model = false, swept = false, state_ready = false;
    int jc = 0;
    std::vector<double> h_lam_l, h_lam_r, h_ldf;
    // model + leaves
    double *d_Q = nullptr, *d_pi = nullptr, *d_lam_l = nullptr, *d_lam_r = nullptr, *d_ldf = nullptr;
    double* d_leaves = nullptr;          // [N][S][4]
    uint8_t* d_leaf_codes = nullptr;     // [N][S]; in use only when every leaf row is one-hot or all-ones
    bool leaves_coded = false;
    uint32_t* d_pair_hist = nullptr;     // [N][N][32] code-pair site counts of coded leaves (built on the first twisted sweep)
    bool hist_ready = false, codes_valid = false;
    // sweep state
    double* d_pool = nullptr;            // [(N-1)][Kloc][S][4]
    double* d_nodell = nullptr;          // [N + (N-1)*K]
    double *d_bl = nullptr, *d_br = nullptr, *d_Pmat = nullptr;   // [(N-1)][Kloc](x32)
    double *d_logw = nullptr, *d_ll = nullptr;                    // [(N-1)][K] (global columns)
    double* d_aux = nullptr;             // [Kloc][PK_AUX]
    int tile_override = 0;               // PHYLO_SITE_TILE / phylo_set_site_tile (0: the policy pm_site_tile)
    int site_tile = 0, ntiles = 1;       // contract v5: sites per tile of the canonical sum over sites (multiple of 64), ceil(S / tile)
    double* d_tilev = nullptr;           // [Kloc][ntiles] tile values of the merge (rows longer than one tile)
    double* d_lse = nullptr;             // [PK_MAX_GROUPS][N-1 + total]; group 0 only unless sweeps are batched
    uint64_t* d_group_seeds = nullptr;   // [PK_MAX_GROUPS]
    // root tables, two planes each, carved from ONE slab (so that peers map it with one handle):
    // rootll[0], rootll[1] (double), roots[0], roots[1], cnt[0], cnt[1] (int32), each [K][N]
    char* d_tables = nullptr;
    int32_t *d_roots[2] = {nullptr, nullptr}, *d_cnt[2] = {nullptr, nullptr};   // [K][N]
    double* d_rootll[2] = {nullptr, nullptr};                                   // [K][N]
    const char** d_tab_ptrs = nullptr;   // [world] table slab of every rank (peer mappings)
    int32_t* d_child = nullptr;          // [(N-1)][Kloc][2]: children of every node (kept for lazy materialisation)
    unsigned int* d_mark = nullptr;      // [(N-1)][K]: node is in the pool
    double* d_sync = nullptr;  // [world] dummy payload of the barrier collective used by lazy nodes when sharded
    // device-side exchange between ranks (pk_p2p_exchange): the all-gathered arrays live in ONE fine-grained slab that the peers map
    bool p2p = false;
    char* d_xslab = nullptr;             // logw | ll | nodell | chosen | sync | flags[2][world]
    size_t xslab_bytes = 0, x_flag_off[2] = {0, 0};
    char** d_xslab_ptrs = nullptr;       // [world] every rank's slab as mapped here
    unsigned long long x_epoch[2] = {0, 0};   // exchanges / barriers issued so far (the same on every rank)
    bool last_lazy = false;
    int32_t* d_merges = nullptr;         // [(N-1)][Kloc][2]
    int64_t* d_anc = nullptr;            // [(N-2)][Kloc]
    uint64_t* d_cdf[2] = {nullptr, nullptr};   // [K], double-buffered across rank events
    unsigned int* d_counter = nullptr;   // [0] scan->bookkeeping flag, [1] hand-off timeout word
    unsigned int epoch = 0;              // monotone hand-off epoch (never reset, never 0)
    const double** d_pool_ptrs = nullptr; // [world] pool base of every rank (peer mappings)
    // sharded: remote nodes merged by this rank, fetched once per sweep (pk_pull_remote_children)
    int32_t* d_mirror = nullptr;         // [(N-1) K + 4]: node -> slot + 1 | 0 | -2; the last four words: [0] slots taken
    double* d_cache = nullptr;           // [cache_cap][S][4]
    int cache_cap = 0;
    // twisted proposal (allocated on first use)
    int32_t *d_roots_ad = nullptr, *d_cnt_ad = nullptr;
    double *d_rootll_ad = nullptr, *d_chosen = nullptr, *d_tw_b = nullptr, *d_tw_P = nullptr, *d_pot = nullptr;
    size_t tw_capacity = 0;              // in (particle, sub-sample) entries
    double* d_twbuf = nullptr;           // [Kloc][J] softmax weights of pk_twist_choose when J exceeds what LDS holds
    size_t twbuf_cap = 0;
    // ... and its history when the graph is kept (PHYLO_TWISTING | PHYLO_KEEP_GRAPH): every rank event's rows
    double *d_htw_b = nullptr, *d_htw_P = nullptr, *d_hpot = nullptr, *d_hchosen = nullptr;   // [rows][2], [rows][32], [rows], [R][K]
    int32_t* d_hroots_ad = nullptr;      // [R][K][N]
    double *d_tau = nullptr, *d_ctw = nullptr, *d_twpart = nullptr, *d_twnode = nullptr;       // reverse pass
    int64_t* d_joff = nullptr;           // [R+1]
    size_t htw_rows = 0;                 // rows the history holds
    std::vector<int64_t> h_joff;
    bool last_graph_twist = false, last_graph_marks = false;   // marks: the sweep was lazy, d_mark says which nodes were adopted
    int last_M = 1;
    // graph kept for the reverse pass (PHYLO_KEEP_GRAPH; allocated on first use)
    int32_t *d_hroots = nullptr, *d_hcnt = nullptr, *d_pos = nullptr;   // [(R+1)][K][N], [(R+1)][K][N], [R][K][N]
    double* d_hrootll = nullptr;         // [(R+1)][K][N]
    double *d_adj = nullptr, *d_om = nullptr, *d_G = nullptr, *d_C = nullptr, *d_part = nullptr, *d_nodeg = nullptr;
    double *d_leafpi = nullptr, *d_leafterm = nullptr, *d_terms = nullptr, *d_gout = nullptr;
    int32_t *d_ad_off = nullptr, *d_ad_idx = nullptr, *d_par_off = nullptr, *d_par_idx = nullptr;
    int32_t *d_heavy = nullptr, *d_chunk_beg = nullptr, *d_chunk_cnt = nullptr;   // [R K], [<= 2 R K / PG_PCHUNK + 1] x2
    int32_t *d_slow_flag = nullptr, *d_slow_idx = nullptr, *d_adp = nullptr;      // [R K] x3
    bool graph_ready = false, last_graph = false;
    int last_G = 1;
    bool last_final_missing = false;
    std::vector<uint64_t> h_group_seeds;
    // host copies of the kept graph's integer records, in pinned memory: copied asynchronously when the sweep ends, so that the
    // reverse pass finds them on the host without a synchronous copy; and the pinned staging area of its packed integer lists
    int64_t* h_anc_p = nullptr;          // [(R-1)][K]
    double* h_model_p = nullptr;         // pinned image of the model upload (phylo_set_model)
    double* h_leaves_p = nullptr;        // pinned image of the leaf rows and their codes (phylo_set_leaves)
    hipEvent_t ev_leaves = nullptr;
    uint32_t *h_pub = nullptr, *hd_pub = nullptr;
    int32_t *h_dlmeta = nullptr, *hd_dlmeta = nullptr;   // what pg_dl_lists tells the host (phylo_revlists_dev.h), pinned
    unsigned int* d_row_done = nullptr;  // [R K][tiles of 256 sites]: pg_nodes_rows_all's "this tile of the adjoint row is complete" words
    unsigned int row_epoch = 0;          // their value in the current reverse pass
    hipEvent_t ev_dl = nullptr;
    size_t dl_temp_p = 0, dl_temp_nn = 0;   // rocPRIM's temporary storage for the parents' sort at this R K
    hipGraphExec_t dl_graph = nullptr;     // that sort's launches, captured (dev_lists_launch)
    const void* dl_graph_key[4] = {nullptr, nullptr, nullptr, nullptr};
    bool dl_no_graph = false;
    uint32_t *hd_csr = nullptr, *hd_anc = nullptr, *hd_child = nullptr, *hd_rad = nullptr;   // device views of h_csr_p, h_anc_p, h_child_p, h_rad_p
    int32_t *h_child_p = nullptr, *h_rad_p = nullptr, *h_csr_p = nullptr;   // [R][K][2], [R][K][N] (twisted), the d_ad_off slab
    size_t h_csr_cap = 0;                // int32 elements
    hipEvent_t ev_gcopy = nullptr;
    std::vector<int32_t> h_cur;          // scratch of the counting sorts
    std::vector<double> h_vi_lam;        // phylo_vi_gradients: the rates of the step (phylo_set_model copies them)
    std::vector<int32_t> h_xlists;       // ... of its twisted part
    hipEvent_t evb0 = nullptr, evb1 = nullptr, ev_model = nullptr;
    // reverse pass: the adopted nodes' chain runs on gstream beside the coefficient chain on `stream`; ev_coeff[r]: C of rank event r done
    hipStream_t gstream = nullptr;
    hipEvent_t ev_gfork = nullptr, ev_gjoin = nullptr, ev_gup = nullptr;
    hipStream_t bgstream = nullptr;      // lowest priority: pg_nodes_free in the background of the chains
    hipEvent_t ev_bgfork = nullptr, ev_bgdone = nullptr;
    std::vector<hipEvent_t> ev_coeff;
    phylo_stats stats{};
    sweep_run run;
    int n_merge_events = 0;
    // grow-only scratch for the op-level entry points
    DevBuf scratch[12];                  // (8..10: the device-built lists of the reverse pass)
    phylo_comm comm;
};

namespace {

int fail(phylo_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_last_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PHYLO_ENOMEM : PHYLO_EHIP, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)

#define CHK(expr)                   \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != PHYLO_OK) return rc_; \
    } while (0)

template <typename T>
int dalloc(phylo_ctx* ctx, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(ctx, hipMalloc((void**)p, count * sizeof(T)));
    return PHYLO_OK;
}

int scratch_get(phylo_ctx* ctx, int slot, size_t bytes, void** out) {
    DevBuf& b = ctx->scratch[slot];
    if (b.bytes < bytes) {
        if (b.p) HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
        const size_t want = bytes + bytes / 2 + 16;         // (grow by half: sizes that creep up from call to call -- the chunk counts
        HIPCHK(ctx, hipMalloc(&b.p, want));                 //  of the reverse pass -- would free and allocate, 60 us, again and again)
        b.bytes = want;
    }
    *out = b.p;
    return PHYLO_OK;
}

int bind(phylo_ctx* ctx) {
    if (!ctx) return fail(nullptr, PHYLO_EINVAL, "ctx is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return PHYLO_OK;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// -log (2 max(c,2) - 3)!! by the reference's loop (vcsmc.py:30-57): n, n-2, ... while >= 2
double host_log_double_factorial(int m) {
    double res = 0.0;
    for (int v = m; v >= 2; v -= 2) res = res + pm_log((double)v);
    return res;
}

int launch_check(phylo_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, PHYLO_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PHYLO_OK;
}

void free_sweep_state(phylo_ctx* c) {
    // the reverse pass's side streams may still read what is freed below
    if (c->gstream) (void)hipStreamSynchronize(c->gstream);
    if (c->bgstream) (void)hipStreamSynchronize(c->bgstream);
    if (c->d_xslab) {                                       // the exchanged arrays are carved from the slab
        (void)hipFree(c->d_xslab);
        c->d_xslab = nullptr;
        c->d_logw = c->d_ll = c->d_nodell = c->d_sync = nullptr;
        c->d_chosen = nullptr;
    }
    if (c->d_xslab_ptrs) (void)hipFree((void*)c->d_xslab_ptrs);
    c->d_xslab_ptrs = nullptr;
    c->xslab_bytes = 0;
    c->x_epoch[0] = c->x_epoch[1] = 0;
    if (c->d_twbuf) (void)hipFree(c->d_twbuf);
    c->d_twbuf = nullptr;
    c->twbuf_cap = 0;
    void* tw[] = {c->d_roots_ad, c->d_cnt_ad, c->d_rootll_ad, c->d_chosen, c->d_tw_b, c->d_tw_P, c->d_pot};
    for (void* p : tw)
        if (p) (void)hipFree(p);
    c->d_roots_ad = c->d_cnt_ad = nullptr;
    c->d_rootll_ad = c->d_chosen = c->d_tw_b = c->d_tw_P = c->d_pot = nullptr;
    c->tw_capacity = 0;
    void* ht[] = {c->d_htw_b, c->d_htw_P, c->d_hpot, c->d_hchosen, c->d_hroots_ad, c->d_tau, c->d_ctw, c->d_twpart, c->d_twnode, c->d_joff};
    for (void* p : ht)
        if (p) (void)hipFree(p);
    c->d_htw_b = c->d_htw_P = c->d_hpot = c->d_hchosen = c->d_tau = c->d_ctw = c->d_twpart = c->d_twnode = nullptr;
    c->d_hroots_ad = nullptr;
    c->d_joff = nullptr;
    c->htw_rows = 0;
    c->last_graph_twist = false;
    void* gr[] = {c->d_hroots, c->d_hcnt, c->d_pos, c->d_hrootll, c->d_adj, c->d_om, c->d_G, c->d_C, c->d_part, c->d_nodeg,
                  c->d_leafpi, c->d_leafterm, c->d_terms, c->d_gout, c->d_ad_off};
    for (void* p : gr)
        if (p) (void)hipFree(p);
    c->d_hroots = c->d_hcnt = c->d_pos = nullptr;
    c->d_hrootll = c->d_adj = c->d_om = c->d_G = c->d_C = c->d_part = c->d_nodeg = nullptr;
    c->d_leafpi = c->d_leafterm = c->d_terms = c->d_gout = nullptr;
    c->d_ad_off = c->d_ad_idx = c->d_par_off = c->d_par_idx = nullptr;
    c->d_heavy = c->d_chunk_beg = c->d_chunk_cnt = nullptr;
    c->d_slow_flag = c->d_slow_idx = c->d_adp = nullptr;
    if (c->h_csr_p) (void)hipHostFree(c->h_csr_p);
    if (c->h_anc_p) (void)hipHostFree(c->h_anc_p);
    if (c->h_child_p) (void)hipHostFree(c->h_child_p);
    if (c->h_rad_p) (void)hipHostFree(c->h_rad_p);
    if (c->h_pub) (void)hipHostFree(c->h_pub);
    if (c->d_row_done) (void)hipFree(c->d_row_done);
    c->d_row_done = nullptr;
    if (c->h_dlmeta) (void)hipHostFree(c->h_dlmeta);
    c->h_dlmeta = c->hd_dlmeta = nullptr;
    if (c->ev_dl) (void)hipEventDestroy(c->ev_dl);
    c->ev_dl = nullptr;
    if (c->dl_graph) (void)hipGraphExecDestroy(c->dl_graph);
    c->dl_graph = nullptr;
    c->h_pub = c->hd_pub = nullptr;
    c->h_csr_p = c->h_child_p = c->h_rad_p = nullptr;
    c->h_anc_p = nullptr;
    if (c->ev_gcopy) (void)hipEventDestroy(c->ev_gcopy);
    c->ev_gcopy = nullptr;
    c->graph_ready = false;
    c->last_graph = false;
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    c->d_stamps = nullptr;
    if (c->d_rdraw) (void)hipFree(c->d_rdraw);
    if (c->d_pctr) (void)hipFree(c->d_pctr);
    c->d_rdraw = c->d_pctr = nullptr;
    c->pctr_base = 0;
    c->pctr_Wg = c->pctr_G = 0;
    c->pctr_dirty = false;
    if (c->d_tilev) (void)hipFree(c->d_tilev);
    c->d_tilev = nullptr;
    void* ptrs[] = {c->d_pool, c->d_nodell, c->d_bl, c->d_br, c->d_Pmat, c->d_logw, c->d_ll, c->d_aux, c->d_lse, c->d_group_seeds,
                    c->d_tables, (void*)c->d_tab_ptrs, c->d_child, c->d_merges, c->d_anc,
                    c->d_cdf[0], c->d_cdf[1], c->d_counter, (void*)c->d_pool_ptrs, c->d_mark, c->d_sync, c->d_mirror, c->d_cache};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->d_pool = c->d_nodell = c->d_bl = c->d_br = c->d_Pmat = c->d_logw = c->d_ll = c->d_aux = c->d_lse = nullptr;
    c->d_group_seeds = nullptr;
    c->d_roots[0] = c->d_roots[1] = c->d_cnt[0] = c->d_cnt[1] = c->d_child = c->d_merges = nullptr;
    c->d_anc = nullptr;
    c->d_cdf[0] = c->d_cdf[1] = nullptr;
    c->d_counter = nullptr;
    c->d_rootll[0] = c->d_rootll[1] = nullptr;
    c->d_tables = nullptr;
    c->d_tab_ptrs = nullptr;
    c->d_pool_ptrs = nullptr;
    c->d_mark = nullptr;
    c->d_sync = nullptr;
    c->d_mirror = nullptr;
    c->d_cache = nullptr;
    c->cache_cap = 0;
}

int alloc_sweep_state(phylo_ctx* c) {
    free_sweep_state(c);
    c->state_ready = false;
    const size_t R = (size_t)c->N - 1, K = c->K, Kl = c->Kloc, N = c->N, S = c->S;
    CHK(dalloc(c, &c->d_pool, R * Kl * S * 4));
    // Sharded: the arrays that cross ranks at every rank event live in one slab of FINE-GRAINED device memory (peers write into
    // it over xGMI and this rank reads it in later kernels: no stale line may survive in this device's L2) that every peer maps.
    c->p2p = c->comm.transport != 0 && c->world > 1 && !c->env.no_p2p && R >= 2;
    if (c->p2p) {
        const size_t al = 32;                               // doubles: 256-byte alignment of every array
        auto up = [&](size_t n) { return (n + al - 1) / al * al; };
        const size_t n_logw = up(R * K), n_nod = up(N + R * K), n_ch = up(K), n_sy = up((size_t)c->world), n_fl = up((size_t)c->world);
        const size_t total = 2 * n_logw + n_nod + n_ch + n_sy + 2 * n_fl;
        void* slab = nullptr;
        if (hipExtMallocWithFlags(&slab, total * 8, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipMalloc(&slab, total * 8));         // (coarse-grained: still correct on one device)
        }
        HIPCHK(c, hipMemset(slab, 0, total * 8));
        c->d_xslab = (char*)slab;
        c->xslab_bytes = total * 8;
        double* base = (double*)slab;
        c->d_logw = base; c->d_ll = base + n_logw; c->d_nodell = base + 2 * n_logw;
        c->d_chosen = c->d_nodell + n_nod;
        c->d_sync = c->d_chosen + n_ch;
        c->x_flag_off[0] = (size_t)((char*)(c->d_sync + n_sy) - c->d_xslab);
        c->x_flag_off[1] = c->x_flag_off[0] + n_fl * 8;
        c->x_epoch[0] = c->x_epoch[1] = 0;
    } else {
    CHK(dalloc(c, &c->d_nodell, N + R * K));
    }
    CHK(dalloc(c, &c->d_bl, R * Kl));
    CHK(dalloc(c, &c->d_br, R * Kl));
    CHK(dalloc(c, &c->d_Pmat, R * Kl * 32));
    if (!c->p2p) {
        CHK(dalloc(c, &c->d_logw, R * K));
        CHK(dalloc(c, &c->d_ll, R * K));
    }
    CHK(dalloc(c, &c->d_aux, Kl * PK_AUX));
    if (c->ntiles > 1) CHK(dalloc(c, &c->d_tilev, Kl * (size_t)c->ntiles));
    CHK(dalloc(c, &c->d_lse, (R + 1) * PK_MAX_GROUPS));
    CHK(dalloc(c, &c->d_group_seeds, PK_MAX_GROUPS));
    CHK(dalloc(c, &c->d_tables, 32 * K * N));
    for (int i = 0; i < 2; ++i) {
        c->d_rootll[i] = reinterpret_cast<double*>(c->d_tables) + (size_t)i * K * N;
        c->d_roots[i] = reinterpret_cast<int32_t*>(c->d_tables + 16 * K * N) + (size_t)i * K * N;
        c->d_cnt[i] = reinterpret_cast<int32_t*>(c->d_tables + 24 * K * N) + (size_t)i * K * N;
    }
    CHK(dalloc(c, &c->d_child, R * Kl * 2));
    CHK(dalloc(c, &c->d_mark, ((R * K + R + 3) & ~(size_t)3)));      // one mark per node
    if (!c->p2p) CHK(dalloc(c, &c->d_sync, (size_t)c->world));
    CHK(dalloc(c, &c->d_merges, R * Kl * 2));
    CHK(dalloc(c, &c->d_anc, (R > 0 ? R - 1 : 0) * Kl));
    CHK(dalloc(c, &c->d_cdf[0], K));
    CHK(dalloc(c, &c->d_cdf[1], K));
    CHK(dalloc(c, &c->d_counter, ((size_t)N + 3) & ~(size_t)3));
    HIPCHK(c, hipMemset(c->d_counter, 0, (((size_t)N + 3) & ~(size_t)3) * sizeof(unsigned int)));
    CHK(dalloc(c, &c->d_pool_ptrs, (size_t)c->world));
    std::vector<void*> ptrs;
    int rc = phylo_comm_map_pools(c->comm, c->d_pool, &ptrs, c->stream, &c->err);
    if (rc != PHYLO_OK) return rc;
    HIPCHK(c, hipMemcpy((void*)c->d_pool_ptrs, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice));
    if ((c->world > 1 || c->env.rehearse_sharded) && !c->env.no_remote_cache) {
        // the local cache of remote nodes: up to 512 MB of rows (primate.p: 17 000 nodes; 128 x 50 000: 320), the rest in place
        const size_t node_bytes = (size_t)S * 32;
        size_t cap = ((size_t)512 << 20) / node_bytes;
        if (c->env.remote_cache_cap > 0) cap = (size_t)c->env.remote_cache_cap;
        if (cap > R * K) cap = R * K;
        if (cap < 1) cap = 1;
        c->cache_cap = (int)cap;
        CHK(dalloc(c, &c->d_mirror, R * K + 4));
        CHK(dalloc(c, &c->d_cache, cap * (size_t)S * 4));
    }
    CHK(dalloc(c, &c->d_tab_ptrs, (size_t)c->world));
    rc = phylo_comm_map_extra(c->comm, c->d_tables, &ptrs, c->stream, &c->err);
    if (rc != PHYLO_OK) return rc;
    HIPCHK(c, hipMemcpy((void*)c->d_tab_ptrs, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice));
    if (c->p2p) {
        CHK(dalloc(c, &c->d_xslab_ptrs, (size_t)c->world));
        rc = phylo_comm_map_extra(c->comm, c->d_xslab, &ptrs, c->stream, &c->err);
        if (rc != PHYLO_OK) return rc;
        HIPCHK(c, hipMemcpy((void*)c->d_xslab_ptrs, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice));
    }
    c->state_ready = true;
    return PHYLO_OK;
}

// The sweep state (node pool = (N-1) K_local S 32 bytes) is allocated on first use, so that a context created
// with the GLOBAL particle count and then sharded by phylo_comm_init never asks for the unsharded pool.
int ensure_sweep_state(phylo_ctx* c) {
    if (c->state_ready) return PHYLO_OK;
    return alloc_sweep_state(c);
}

// buffers of the reverse pass: table history, adjoint pool, coefficient tables, CSR lists
int ensure_graph_state(phylo_ctx* c) {
    if (c->graph_ready) return PHYLO_OK;
    const size_t R = (size_t)c->N - 1, K = c->K, N = c->N, S = c->S;
    const size_t T = (S + PG_NT - 1) / PG_NT;
    CHK(dalloc(c, &c->d_hroots, (R + 1) * K * N));
    CHK(dalloc(c, &c->d_hcnt, (R + 1) * K * N));
    CHK(dalloc(c, &c->d_hrootll, (R + 1) * K * N));
    CHK(dalloc(c, &c->d_pos, R * K * N));
    CHK(dalloc(c, &c->d_adj, R * K * S * 4));
    CHK(dalloc(c, &c->d_om, R * K));
    CHK(dalloc(c, &c->d_G, R * K));
    CHK(dalloc(c, &c->d_C, R * K * N));
    CHK(dalloc(c, &c->d_part, R * K * T * PG_PART));
    CHK(dalloc(c, &c->d_nodeg, R * K * PG_NODEG));
    CHK(dalloc(c, &c->d_leafpi, N * 4));
    CHK(dalloc(c, &c->d_leafterm, K * 4));
    CHK(dalloc(c, &c->d_terms, R * K * 2));
    CHK(dalloc(c, &c->d_gout, 2 * R + 20));
    {   // the integer lists of the reverse pass live in ONE slab, uploaded with one copy per step
        c->h_csr_cap = pg_lists_ints(R, K);               // (layout: pg_lists_carve, phylo_revlists.h)
        CHK(dalloc(c, &c->d_ad_off, c->h_csr_cap));
        HIPCHK(c, hipHostMalloc((void**)&c->h_csr_p, c->h_csr_cap * 4));
        HIPCHK(c, hipHostMalloc((void**)&c->h_anc_p, (R > 1 ? (R - 1) * K : 1) * 8));
        HIPCHK(c, hipHostMalloc((void**)&c->h_child_p, R * K * 2 * 4));
        // device views of the pinned buffers (pg_copy_words reads / writes them from kernels)
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_csr, c->h_csr_p, 0));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_anc, c->h_anc_p, 0));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_child, c->h_child_p, 0));
        HIPCHK(c, hipHostMalloc((void**)&c->h_pub, 16));   // log Z-hat (8 bytes) and the timeout word of a sweep that keeps its graph
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_pub, c->h_pub, 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_gcopy, hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void**)&c->h_dlmeta, ((size_t)PG_DL_META_INTS(R) + 4) * 4));   // (+ the timeout word of pg_nodes_rows_all)
        c->h_dlmeta[PG_DL_META_INTS(R)] = 0;
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_dlmeta, c->h_dlmeta, 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_dl, hipEventDisableTiming));
        c->dl_temp_nn = 0;
        const pg_lists D = pg_lists_carve(c->d_ad_off, R, K);
        c->d_ad_idx = D.ad_idx; c->d_par_off = D.par_off; c->d_par_idx = D.par_idx;
        c->d_heavy = D.heavy; c->d_chunk_beg = D.chunk_beg; c->d_chunk_cnt = D.chunk_cnt;
        c->d_slow_flag = D.slow_flag; c->d_slow_idx = D.slow_idx; c->d_adp = D.adp;
    }
    if (!c->evb0) {
        HIPCHK(c, hipEventCreate(&c->evb0));
        HIPCHK(c, hipEventCreate(&c->evb1));
        HIPCHK(c, hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_gfork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_gjoin, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_gup, hipEventDisableTiming));
        {
            int least = 0, greatest = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(c, hipStreamCreateWithPriority(&c->bgstream, hipStreamNonBlocking, least));
        }
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_bgfork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_bgdone, hipEventDisableTiming));
        c->ev_coeff.assign((size_t)R, nullptr);
        for (auto& e : c->ev_coeff) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    c->graph_ready = true;
    return PHYLO_OK;
}

// resampling scan of G groups of Kg log-weights: the LDS form when a group fits (phylo_persist.h), else pk_resample_scan
int launch_scan(phylo_ctx* c, const double* logw, int Kg, int G, uint64_t* cdf, double* lse, int lse_stride, int logz_R = 0) {
    if (Kg > c->env.scan_multi_min && (cdf || lse)) {      // large groups: several workgroups per group, three small launches
        pp_scan_multi_args a{};
        a.logw = logw; a.Kg = Kg; a.B = (Kg + PP_SCAN_TILE - 1) / PP_SCAN_TILE;
        void* ws = nullptr;
        const size_t nb = (size_t)G * a.B;
        CHK(scratch_get(c, 11, (nb * 2 + (size_t)G * Kg) * 8, &ws));
        a.gmax = (double*)ws; a.bsum = (unsigned long long*)ws + nb; a.wbits = (unsigned long long*)ws + 2 * nb;
        a.cdf = (unsigned long long*)cdf; a.lse_out = lse; a.lse_stride = lse_stride; a.logz_R = logz_R;
        hipLaunchKernelGGL(pp_scan_multi_max, dim3(a.B, G), dim3(512), 0, c->stream, a);
        CHK(launch_check(c, "pp_scan_multi_max"));
        hipLaunchKernelGGL(pp_scan_multi_exp, dim3(a.B, G), dim3(512), 0, c->stream, a);
        CHK(launch_check(c, "pp_scan_multi_exp"));
        // (no cdf wanted: the log-normaliser's workgroup alone)
        if (cdf) hipLaunchKernelGGL(pp_scan_multi_cdf, dim3(a.B + 1, G), dim3(512), 0, c->stream, a);
        else { a.lse_only = 1; hipLaunchKernelGGL(pp_scan_multi_cdf, dim3(1, G), dim3(512), 0, c->stream, a); }
        return launch_check(c, "pp_scan_multi_cdf");
    }
    if (Kg <= 4096) {
        hipLaunchKernelGGL(pp_resample_scan<512>, dim3(G), dim3(512), pp_resample_scan_lds(Kg), c->stream, logw, Kg, cdf, lse, lse_stride, logz_R);
        return launch_check(c, "pp_resample_scan");
    }
    if (Kg <= PP_SCAN_KERNEL_MAX_KG) {      // large groups (the replicated scan of a sharded sweep): 16 waves, 128 KiB of LDS
        hipLaunchKernelGGL(pp_resample_scan<1024>, dim3(G), dim3(1024), pp_resample_scan_lds(Kg), c->stream, logw, Kg, cdf, lse, lse_stride, logz_R);
        return launch_check(c, "pp_resample_scan");
    }
    if (G > 1) hipLaunchKernelGGL(pk_resample_scan_groups, dim3(G), dim3(PK_COLS), pk_scan_lds_bytes(Kg), c->stream, logw, Kg, cdf, lse, lse_stride);
    else hipLaunchKernelGGL(pk_resample_scan, dim3(1), dim3(PK_COLS), pk_scan_lds_bytes(Kg), c->stream, logw, Kg, cdf, lse);
    return launch_check(c, "pk_resample_scan");
}

// leaf node log-likelihoods sum_s log(pi . leaf[s]) (depend on pi and the leaves)
// Wait for an event by polling (a blocking hipEventSynchronize wakes the thread tens of microseconds after the event -- a twentieth of
// a training step, twice per step); after 2 ms of polling, block.
int wait_event_spin(phylo_ctx* c, hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (; !c->env.no_spin_wait;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return PHYLO_OK;
        if (e != hipErrorNotReady) return fail(c, PHYLO_EHIP, "hipEventQuery: %s", hipGetErrorString(e));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        __builtin_ia32_pause();
    }
    HIPCHK(c, hipEventSynchronize(ev));
    return PHYLO_OK;
}

// site tile of an op-level call on rows of S sites (the context's own S has c->site_tile)
int tile_for(const phylo_ctx* c, int S) { return c->tile_override ? c->tile_override : pm_site_tile(S); }

int refresh_leaf_ll(phylo_ctx* c) {
    if (!(c->have_leaves && c->have_model && c->state_ready)) return PHYLO_OK;
    hipLaunchKernelGGL(pk_row_loglik, dim3(c->N), dim3(64), 0, c->stream, c->d_leaves, c->d_pi, c->S, c->site_tile,
                       c->d_nodell);
    return launch_check(c, "pk_row_loglik(leaves)");
}

}  // namespace

extern "C" {

const char* phylo_version(void) { return "phylo_hip 0.1 (gfx950)"; }

const char* phylo_last_error(const phylo_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int phylo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int phylo_create(const int* device_ids, int n_gpus, int K, int N, int S, int A, uint32_t flags,
                 phylo_ctx** out) {
    if (!out) return fail(nullptr, PHYLO_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_gpus != 1)
        return fail(nullptr, PHYLO_EINVAL, "n_gpus must be 1 (one process per GPU; join ranks with phylo_comm_init)");
    if (A != 4) return fail(nullptr, PHYLO_EINVAL, "A must be 4 (DNA alphabet), got %d", A);
    if (K < 1 || N < 2 || S < 1) return fail(nullptr, PHYLO_EINVAL, "need K >= 1, N >= 2, S >= 1 (K=%d N=%d S=%d)", K, N, S);
    if (N > PK_MAX_TAXA) return fail(nullptr, PHYLO_EINVAL, "N = %d exceeds the supported maximum %d", N, PK_MAX_TAXA);
    if ((double)(N - 1) * K + N > 2.0e9) return fail(nullptr, PHYLO_EINVAL, "node ids overflow int32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, PHYLO_ENODEVICE, "no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    const int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= ndev) return fail(nullptr, PHYLO_ENODEVICE, "device id %d out of range [0,%d)", dev, ndev);
    phylo_ctx* c = new phylo_ctx();
    c->device = dev;
    c->K = K; c->N = N; c->S = S; c->A = A;
    c->Kloc = K; c->k0 = 0;
    c->flags = flags;
    c->env.read();
    {   // contract v5: the site tile is part of the arithmetic contract (the oracle takes the same value)
        const char* t = getenv("PHYLO_SITE_TILE");
        int T = t ? atoi(t) : pm_site_tile(S);
        if (T < 64 || (T & 63) || T > PK_MAX_SITE_TILE) { delete c; return fail(nullptr, PHYLO_EINVAL, "PHYLO_SITE_TILE must be a multiple of 64 in [64, %d] (got %d)", PK_MAX_SITE_TILE, T); }
        c->tile_override = t ? T : 0;
        c->site_tile = T;
        c->ntiles = (S + T - 1) / T;
    }
    int rc = PHYLO_OK;
    do {
        if ((rc = bind(c)) != PHYLO_OK) break;
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { rc = fail(c, PHYLO_EHIP, "hipGetDeviceProperties failed"); break; }
            c->n_cus = prop.multiProcessorCount;
        }
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(c, PHYLO_EHIP, "hipStreamCreate failed"); break; }
        if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { rc = fail(c, PHYLO_EHIP, "hipEventCreate failed"); break; }
        if ((rc = dalloc(c, &c->d_Q, 20 + 2 * (size_t)N)) != PHYLO_OK) break;   // one slab: Q[16] pi[4] lam_l[N] lam_r[N]
        c->d_pi = c->d_Q + 16;
        c->d_lam_l = c->d_Q + 20;
        c->d_lam_r = c->d_Q + 20 + N;
        if ((rc = dalloc(c, &c->d_ldf, (size_t)N + 1)) != PHYLO_OK) break;
        if ((rc = dalloc(c, &c->d_leaves, (size_t)N * S * 4)) != PHYLO_OK) break;
        if ((rc = dalloc(c, &c->d_leaf_codes, (size_t)N * S)) != PHYLO_OK) break;
        // table of log (2 max(c,2) - 3)!! by leaf count c = 0..N
        c->h_ldf.resize((size_t)N + 1);
        for (int cnt = 0; cnt <= N; ++cnt) c->h_ldf[cnt] = host_log_double_factorial(2 * (cnt > 2 ? cnt : 2) - 3);
        if (hipMemcpy(c->d_ldf, c->h_ldf.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(c, PHYLO_EHIP, "ldf upload failed"); break; }
    } while (0);
    if (rc != PHYLO_OK) {
        g_last_error = c->err;
        phylo_destroy(c);
        return rc;
    }
    *out = c;
    return PHYLO_OK;
}

int phylo_destroy(phylo_ctx* c) {
    if (!c) return PHYLO_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->gstream) (void)hipStreamSynchronize(c->gstream);
    if (c->bgstream) (void)hipStreamSynchronize(c->bgstream);
    phylo_comm_destroy(&c->comm);
    free_sweep_state(c);
    void* ptrs[] = {c->d_Q, c->d_ldf, c->d_leaves, c->d_leaf_codes};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& b : c->scratch)
        if (b.p) (void)hipFree(b.p);
    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evb0) (void)hipEventDestroy(c->evb0);
    if (c->evb1) (void)hipEventDestroy(c->evb1);
    if (c->ev_model) (void)hipEventDestroy(c->ev_model);
    if (c->ev_gfork) (void)hipEventDestroy(c->ev_gfork);
    if (c->ev_gjoin) (void)hipEventDestroy(c->ev_gjoin);
    if (c->ev_gup) (void)hipEventDestroy(c->ev_gup);
    if (c->ev_bgfork) (void)hipEventDestroy(c->ev_bgfork);
    if (c->ev_bgdone) (void)hipEventDestroy(c->ev_bgdone);
    if (c->bgstream) (void)hipStreamDestroy(c->bgstream);
    for (hipEvent_t e : c->ev_coeff)
        if (e) (void)hipEventDestroy(e);
    if (c->gstream) (void)hipStreamDestroy(c->gstream);
    if (c->h_model_p) (void)hipHostFree(c->h_model_p);
    if (c->ev_leaves) (void)hipEventDestroy(c->ev_leaves);
    if (c->h_leaves_p) (void)hipHostFree(c->h_leaves_p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PHYLO_OK;
}

int phylo_site_tile(int S) { return pm_site_tile(S); }

int phylo_get_site_tile(const phylo_ctx* c) { return c ? c->site_tile : 0; }

int phylo_set_site_tile(phylo_ctx* c, int T) {
    CHK(bind(c));
    if (T < 0 || (T & 63) || T > PK_MAX_SITE_TILE)
        return fail(c, PHYLO_EINVAL, "the site tile must be a multiple of 64 in [64, %d], or 0 for the default (got %d)", PK_MAX_SITE_TILE, T);
    if (c->comm.transport != 0) return fail(c, PHYLO_ESTATE, "phylo_set_site_tile must precede phylo_comm_init (peers map the sweep state)");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tile_override = T;
    c->site_tile = T ? T : pm_site_tile(c->S);
    c->ntiles = (c->S + c->site_tile - 1) / c->site_tile;
    free_sweep_state(c);                                   // tile values, leaf log-likelihoods: rebuilt by the next sweep
    c->state_ready = false;
    c->swept = false;
    return PHYLO_OK;
}

int phylo_set_leaves(phylo_ctx* c, const double* genome) {
    CHK(bind(c));
    if (!genome) return fail(c, PHYLO_EINVAL, "genome_NxSxA is NULL");
    // The rows and their 1-byte codes go up from a pinned image that outlives the call, on the context's stream: the call does not
    // wait for the device (a training step on site minibatches sets new leaves every time).  Only a previous upload still in
    // flight has to be over before the image is overwritten.
    const size_t rows = (size_t)c->N * c->S;
    const bool pinned = rows * 33 <= ((size_t)8 << 20);     // (a large alignment goes up straight from the caller's buffer, and waits)
    std::vector<uint8_t> codes_v;
    uint8_t* codes = nullptr;
    if (pinned) {
        if (!c->h_leaves_p) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_leaves_p, rows * 33));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_leaves, hipEventDisableTiming));
        } else {
            HIPCHK(c, hipEventSynchronize(c->ev_leaves));
        }
        memcpy(c->h_leaves_p, genome, rows * 32);
        HIPCHK(c, hipMemcpyAsync(c->d_leaves, c->h_leaves_p, rows * 32, hipMemcpyHostToDevice, c->stream));
        codes = (uint8_t*)c->h_leaves_p + rows * 32;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_leaves, genome, rows * 32, hipMemcpyHostToDevice, c->stream));
        codes_v.resize(rows);
        codes = codes_v.data();
    }
    // one-hot / all-ones rows (the reference's encoding, runner.py:83-96) also get a 1-byte code per site
    {
        bool ok = true;
        for (size_t i = 0; i < rows && ok; ++i) {
            const double* x = genome + i * 4;
            int ones = 0, zeros = 0, last = 0;
            for (int j = 0; j < 4; ++j) {
                if (pm_bits(x[j]) == pm_bits(1.0)) { ++ones; last = j; }
                else if (pm_bits(x[j]) == 0) ++zeros;
            }
            if (ones == 1 && zeros == 3) codes[i] = (uint8_t)last;
            else if (ones == 4) codes[i] = 4;
            else ok = false;
        }
        c->codes_valid = ok;                                  // a property of the data (the twisting contract uses it)
        c->leaves_coded = ok && !c->env.no_leaf_codes;   // the access-path optimisation can be switched off
        c->hist_ready = false;
        if (ok) HIPCHK(c, hipMemcpyAsync(c->d_leaf_codes, codes, rows, hipMemcpyHostToDevice, c->stream));
    }
    if (pinned) HIPCHK(c, hipEventRecord(c->ev_leaves, c->stream));
    c->have_leaves = true;
    c->last_graph = c->last_graph_twist = false;           // ... and to the leaves
    CHK(refresh_leaf_ll(c));
    if (!pinned) HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_set_model(phylo_ctx* c, const double* Q16, const double* pi4, const double* lam_l, const double* lam_r,
                    int jc69_closed_form) {
    CHK(bind(c));
    if (!Q16 || !pi4 || !lam_l || !lam_r) return fail(c, PHYLO_EINVAL, "NULL model pointer");
    const int R = c->N - 1;
    for (int i = 0; i < R; ++i)
        if (!(lam_l[i] > 0.0) || !(lam_r[i] > 0.0)) return fail(c, PHYLO_EINVAL, "branch rates must be positive");
    c->h_lam_l.assign(lam_l, lam_l + R);
    c->h_lam_r.assign(lam_r, lam_r + R);
    c->jc = jc69_closed_form ? 1 : 0;
    // one upload for the 42 numbers, from a pinned image that outlives the call: everything that uses the model is ordered
    // behind it on the context's stream, so the call does not wait (a training step sets a new model every time).  Only a
    // previous upload still in flight has to be over before the image is overwritten.
    const size_t npack = 20 + 2 * (size_t)c->N;
    if (!c->h_model_p) {
        HIPCHK(c, hipHostMalloc((void**)&c->h_model_p, npack * 8));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_model, hipEventDisableTiming));
    } else {
        HIPCHK(c, hipEventSynchronize(c->ev_model));
    }
    double* pack = c->h_model_p;
    memset(pack, 0, npack * 8);
    memcpy(pack, Q16, 16 * 8);
    memcpy(pack + 16, pi4, 4 * 8);
    memcpy(pack + 20, lam_l, (size_t)R * 8);
    memcpy(pack + 20 + c->N, lam_r, (size_t)R * 8);
    HIPCHK(c, hipMemcpyAsync(c->d_Q, pack, npack * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_model, c->stream));
    c->have_model = true;
    c->last_graph = c->last_graph_twist = false;           // a kept graph belongs to the model it was swept with
    CHK(refresh_leaf_ll(c));
    return PHYLO_OK;
}

int phylo_expm_batched(phylo_ctx* c, const double* t, int n, double* P) {
    CHK(bind(c));
    if (!c->have_model) return fail(c, PHYLO_ESTATE, "phylo_set_model has not been called");
    if (n < 0 || (n > 0 && (!t || !P))) return fail(c, PHYLO_EINVAL, "bad arguments to phylo_expm_batched");
    if (n == 0) return PHYLO_OK;
    void *dt, *dP;
    CHK(scratch_get(c, 0, (size_t)n * 8, &dt));
    CHK(scratch_get(c, 1, (size_t)n * 16 * 8, &dP));
    HIPCHK(c, hipMemcpyAsync(dt, t, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pk_expm_batched, dim3(cdiv(n, 64)), dim3(64), 0, c->stream, c->d_Q, (const double*)dt, n, c->jc,
                       (double*)dP);
    CHK(launch_check(c, "pk_expm_batched"));
    HIPCHK(c, hipMemcpyAsync(P, dP, (size_t)n * 16 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_cond_likelihood_K(phylo_ctx* c, const double* l, const double* r, const double* tl, const double* tr, int K,
                            int S, double* out) {
    CHK(bind(c));
    if (!c->have_model) return fail(c, PHYLO_ESTATE, "phylo_set_model has not been called");
    if (K < 0 || S < 0) return fail(c, PHYLO_EINVAL, "negative shape");
    if (K == 0 || S == 0) return PHYLO_OK;
    if (!l || !r || !tl || !tr || !out) return fail(c, PHYLO_EINVAL, "NULL pointer");
    const size_t nb = (size_t)K * S * 4 * 8;
    void *dl, *dr, *dt, *dP, *dout;
    CHK(scratch_get(c, 0, nb, &dl));
    CHK(scratch_get(c, 1, nb, &dr));
    CHK(scratch_get(c, 2, (size_t)2 * K * 8, &dt));
    CHK(scratch_get(c, 3, (size_t)2 * K * 16 * 8, &dP));
    CHK(scratch_get(c, 4, nb, &dout));
    HIPCHK(c, hipMemcpyAsync(dl, l, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dr, r, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dt, tl, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((double*)dt + K, tr, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pk_expm_batched, dim3(cdiv(2 * K, 64)), dim3(64), 0, c->stream, c->d_Q, (const double*)dt, 2 * K,
                       c->jc, (double*)dP);
    CHK(launch_check(c, "pk_expm_batched"));
    hipLaunchKernelGGL(pk_merge_api, dim3(K), dim3(PK_COLS), 0, c->stream, (const double*)dl, (const double*)dr,
                       (const double*)dP, K, S, (double*)dout);
    CHK(launch_check(c, "pk_merge_api"));
    HIPCHK(c, hipMemcpyAsync(out, dout, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_forest_loglik(phylo_ctx* c, const double* core, const int32_t* record, int K, int X, int S, double* out) {
    CHK(bind(c));
    if (!c->have_model) return fail(c, PHYLO_ESTATE, "phylo_set_model has not been called");
    if (K < 0 || X < 0 || S < 0) return fail(c, PHYLO_EINVAL, "negative shape");
    if (K == 0) return PHYLO_OK;
    if (!out || (X > 0 && (!core || !record))) return fail(c, PHYLO_EINVAL, "NULL pointer");
    const size_t rows = (size_t)K * X;
    void *dcore, *drec, *drow, *dout;
    CHK(scratch_get(c, 0, rows * S * 4 * 8, &dcore));
    CHK(scratch_get(c, 1, rows * 4, &drec));
    CHK(scratch_get(c, 2, rows * 8, &drow));
    CHK(scratch_get(c, 3, (size_t)K * 8, &dout));
    if (rows) {
        HIPCHK(c, hipMemcpyAsync(dcore, core, rows * S * 4 * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(drec, record, rows * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pk_row_loglik, dim3((unsigned)rows), dim3(64), 0, c->stream, (const double*)dcore, c->d_pi, S, tile_for(c, S),
                           (double*)drow);
        CHK(launch_check(c, "pk_row_loglik"));
    }
    hipLaunchKernelGGL(pk_forest_tail, dim3(cdiv(K, 256)), dim3(256), 0, c->stream, (const double*)drow,
                       (const int32_t*)drec, c->d_ldf, c->N, K, X, (double*)dout);
    CHK(launch_check(c, "pk_forest_tail"));
    HIPCHK(c, hipMemcpyAsync(out, dout, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_tree_loglik(phylo_ctx* c, int n_nodes, int n_leaves, int S, const int32_t* left, const int32_t* right,
                      const double* bl, const double* br, int root, const double* leaves, const double* prior4,
                      double* out_loglik, double* root_data) {
    CHK(bind(c));
    if (!c->have_model) return fail(c, PHYLO_ESTATE, "phylo_set_model has not been called (Q is needed)");
    if (n_leaves < 1 || n_nodes < n_leaves || S < 1 || root < 0 || root >= n_nodes)
        return fail(c, PHYLO_EINVAL, "bad tree sizes (n_nodes=%d n_leaves=%d S=%d root=%d)", n_nodes, n_leaves, S, root);
    if (!leaves || !prior4 || !out_loglik || (n_nodes > n_leaves && (!left || !right || !bl || !br)))
        return fail(c, PHYLO_EINVAL, "NULL pointer");
    // children-before-parents order of the internal nodes reachable from root (csmc.py:259-298)
    std::vector<int32_t> order;
    std::vector<double> ts;
    {
        std::vector<char> state((size_t)n_nodes, 0);
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int v = stack.back();
            if (v < n_leaves) { stack.pop_back(); continue; }
            const int lc = left[v], rc = right[v];
            if (lc < 0 || lc >= n_nodes || rc < 0 || rc >= n_nodes || lc == v || rc == v)
                return fail(c, PHYLO_EINVAL, "node %d has invalid children (%d, %d)", v, lc, rc);
            if (state[v] == 0) {
                state[v] = 1;
                stack.push_back(lc);
                stack.push_back(rc);
            } else {
                stack.pop_back();
                if (state[v] == 1) {
                    state[v] = 2;
                    order.push_back(v); order.push_back(lc); order.push_back(rc);
                    ts.push_back(bl[v]); ts.push_back(br[v]);
                }
            }
            if ((int)stack.size() > 4 * n_nodes + 8) return fail(c, PHYLO_EINVAL, "tree contains a cycle");
        }
    }
    const int n_int = (int)(order.size() / 3);
    void *dnodes, *dorder, *dt, *dP, *dpr, *dout;
    CHK(scratch_get(c, 0, (size_t)n_nodes * S * 4 * 8, &dnodes));
    CHK(scratch_get(c, 1, (size_t)(n_int ? n_int : 1) * 3 * 4, &dorder));
    CHK(scratch_get(c, 2, (size_t)(n_int ? n_int : 1) * 2 * 8, &dt));
    CHK(scratch_get(c, 3, (size_t)(n_int ? n_int : 1) * 2 * 16 * 8, &dP));
    CHK(scratch_get(c, 4, 4 * 8, &dpr));
    CHK(scratch_get(c, 5, 8, &dout));
    HIPCHK(c, hipMemcpyAsync(dnodes, leaves, (size_t)n_leaves * S * 4 * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dpr, prior4, 4 * 8, hipMemcpyHostToDevice, c->stream));
    if (n_int) {
        HIPCHK(c, hipMemcpyAsync(dorder, order.data(), order.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dt, ts.data(), ts.size() * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pk_expm_batched, dim3(cdiv(2 * n_int, 64)), dim3(64), 0, c->stream, c->d_Q, (const double*)dt,
                           2 * n_int, c->jc, (double*)dP);
        CHK(launch_check(c, "pk_expm_batched"));
        hipLaunchKernelGGL(pk_tree_prune, dim3(cdiv(S, 256)), dim3(256), 0, c->stream, (double*)dnodes,
                           (const int32_t*)dorder, n_int, (const double*)dP, S);
        CHK(launch_check(c, "pk_tree_prune"));
    }
    const double* droot = (const double*)dnodes + (size_t)root * S * 4;
    hipLaunchKernelGGL(pk_row_loglik, dim3(1), dim3(64), 0, c->stream, droot, (const double*)dpr, S, tile_for(c, S), (double*)dout);
    CHK(launch_check(c, "pk_row_loglik"));
    HIPCHK(c, hipMemcpyAsync(out_loglik, dout, 8, hipMemcpyDeviceToHost, c->stream));
    if (root_data) HIPCHK(c, hipMemcpyAsync(root_data, droot, (size_t)S * 4 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_resample(phylo_ctx* c, const double* logw, int K, uint64_t seed, uint32_t step, int64_t* idx) {
    CHK(bind(c));
    if (K < 0) return fail(c, PHYLO_EINVAL, "negative K");
    if (K == 0) return PHYLO_OK;
    if (!logw || !idx) return fail(c, PHYLO_EINVAL, "NULL pointer");
    void *dw, *dcdf, *didx;
    CHK(scratch_get(c, 0, (size_t)K * 8, &dw));
    CHK(scratch_get(c, 1, (size_t)K * 8, &dcdf));
    CHK(scratch_get(c, 2, (size_t)K * 8, &didx));
    HIPCHK(c, hipMemcpyAsync(dw, logw, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    CHK(launch_scan(c, (const double*)dw, K, 1, (uint64_t*)dcdf, (double*)nullptr, 0));
    hipLaunchKernelGGL(pk_resample_search, dim3(cdiv(K, 256)), dim3(256), 0, c->stream, (const uint64_t*)dcdf, K, K, 0, seed,
                       step, (int64_t*)didx);
    CHK(launch_check(c, "pk_resample_search"));
    HIPCHK(c, hipMemcpyAsync(idx, didx, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_log_zsmc(phylo_ctx* c, const double* logw, int R, int K, double* out) {
    CHK(bind(c));
    if (R < 0 || K < 1 || !out || (R > 0 && !logw)) return fail(c, PHYLO_EINVAL, "bad arguments to phylo_log_zsmc");
    void *dw, *dlse;
    CHK(scratch_get(c, 0, (size_t)(R ? R : 1) * K * 8, &dw));
    CHK(scratch_get(c, 1, (size_t)(R + 1) * 8, &dlse));
    if (R) HIPCHK(c, hipMemcpyAsync(dw, logw, (size_t)R * K * 8, hipMemcpyHostToDevice, c->stream));
    for (int r = 0; r < R; ++r) CHK(launch_scan(c, (const double*)dw + (size_t)r * K, K, 1, (uint64_t*)nullptr, (double*)dlse + r, 0));
    hipLaunchKernelGGL(pk_logz_total, dim3(1), dim3(64), 0, c->stream, (const double*)dlse, R, (double*)dlse + R);
    CHK(launch_check(c, "pk_logz_total"));
    HIPCHK(c, hipMemcpyAsync(out, (double*)dlse + R, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

static int sweep_begin_impl(phylo_ctx* c, uint64_t seed, uint32_t flags, int M, const uint64_t* group_seeds, int G) {
    CHK(bind(c));
    c->run.active = false;
    if (G < 1 || G > PK_MAX_GROUPS || c->K % G != 0)
        return fail(c, PHYLO_EINVAL, "a batch needs 1 <= G <= %d sweeps and K = %d divisible by G (got %d)", PK_MAX_GROUPS, c->K, G);
    if (G > 1 && (flags & (PHYLO_TWISTING | PHYLO_KEEP_GRAPH)))
        return fail(c, PHYLO_EINVAL, "batched sweeps need the plain proposal without PHYLO_KEEP_GRAPH");
    if (!c->have_leaves || !c->have_model)
        return fail(c, PHYLO_ESTATE, "phylo_set_leaves and phylo_set_model must be called before a sweep");
    if (!c->state_ready) {
        CHK(ensure_sweep_state(c));
        CHK(refresh_leaf_ll(c));
    }
    const int N = c->N, K = c->K, Kl = c->Kloc, S = c->S, R = N - 1;
    const bool twist = (flags & PHYLO_TWISTING) != 0;
    if (twist) {
        if (M < 1 || M > PK_TWIST_MAX_M) return fail(c, PHYLO_EINVAL, "twisting needs 1 <= M <= %d (got %d)", PK_TWIST_MAX_M, M);
        const size_t Jmax = (size_t)(N * (N - 1) / 2) * M;
        if (Jmax > PK_TWIST_MAX_J) return fail(c, PHYLO_EINVAL, "twisting: C(N,2)*M = %zu exceeds %d", Jmax, PK_TWIST_MAX_J);
        if (Jmax > PK_TWIST_LDS_J && c->twbuf_cap < (size_t)Kl * Jmax) {     // weights of more sub-samples than LDS holds
            if (c->d_twbuf) (void)hipFree(c->d_twbuf);
            c->d_twbuf = nullptr;
            c->twbuf_cap = 0;
            CHK(dalloc(c, &c->d_twbuf, (size_t)Kl * Jmax));
            c->twbuf_cap = (size_t)Kl * Jmax;
        }
        if (!c->d_roots_ad) {
            CHK(dalloc(c, &c->d_roots_ad, (size_t)K * N));
            CHK(dalloc(c, &c->d_cnt_ad, (size_t)K * N));
            CHK(dalloc(c, &c->d_rootll_ad, (size_t)K * N));
            if (!c->p2p) CHK(dalloc(c, &c->d_chosen, (size_t)K));
        }
        if (c->tw_capacity < (size_t)Kl * Jmax) {
            if (c->d_tw_b) { (void)hipFree(c->d_tw_b); (void)hipFree(c->d_tw_P); (void)hipFree(c->d_pot); }
            c->d_tw_b = c->d_tw_P = c->d_pot = nullptr;
            c->tw_capacity = 0;
            CHK(dalloc(c, &c->d_tw_b, (size_t)Kl * Jmax * 2));
            CHK(dalloc(c, &c->d_tw_P, (size_t)Kl * Jmax * 32));
            CHK(dalloc(c, &c->d_pot, (size_t)Kl * Jmax));
            c->tw_capacity = (size_t)Kl * Jmax;
        }
    }
    if (twist && c->codes_valid && !c->hist_ready) {
        if (!c->d_pair_hist) CHK(dalloc(c, &c->d_pair_hist, (size_t)N * N * 32));
        hipLaunchKernelGGL(pk_pair_hist, dim3(N, N), dim3(256), 0, c->stream, (const uint8_t*)c->d_leaf_codes, N, S, c->d_pair_hist);
        CHK(launch_check(c, "pk_pair_hist"));
        c->hist_ready = true;
    }
    const bool timek = (flags & PHYLO_TIME_KERNELS) != 0;
    if (timek && (int)c->kev.size() < 2 * R) {
        while ((int)c->kev.size() < 2 * R) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreate(&e));
            c->kev.push_back(e);
        }
    }
    // lazy nodes: dead stores are most of the HBM traffic of the plain sweep (a node is read again only if its
    // creator survives the next resampling).  Needs every reader on this GPU and the plain proposal.
    // Pays when a node is large (HBM-bound merges); on small nodes the extra launch costs more than the stores.
    const bool graph = (flags & PHYLO_KEEP_GRAPH) != 0;
    if (graph) {
        if (c->world != 1) return fail(c, PHYLO_EINVAL, "PHYLO_KEEP_GRAPH needs an unsharded context");
        CHK(ensure_graph_state(c));
        if (twist) {                                       // every rank event keeps its sub-samples: rows [r][k][J_r]
            c->h_joff.assign((size_t)R + 1, 0);
            for (int r = 0; r < R; ++r) c->h_joff[r + 1] = c->h_joff[r] + (int64_t)K * (((N - r) * (N - r - 1)) / 2) * M;
            const size_t rows = (size_t)c->h_joff[R];
            if (c->htw_rows < rows) {
                void* old[] = {c->d_htw_b, c->d_htw_P, c->d_hpot, c->d_tau, c->d_twpart};
                for (void* p : old)
                    if (p) (void)hipFree(p);
                c->d_htw_b = c->d_htw_P = c->d_hpot = c->d_tau = c->d_twpart = nullptr;
                c->htw_rows = 0;
                CHK(dalloc(c, &c->d_htw_b, rows * 2));
                CHK(dalloc(c, &c->d_htw_P, rows * 32));
                CHK(dalloc(c, &c->d_hpot, rows));
                CHK(dalloc(c, &c->d_tau, rows));
                CHK(dalloc(c, &c->d_twpart, rows * PG_PART));
                c->htw_rows = rows;
            }
            if (!c->d_hroots_ad) {
                CHK(dalloc(c, &c->d_hroots_ad, (size_t)R * K * N));
                CHK(dalloc(c, &c->d_hchosen, (size_t)R * K));
                CHK(dalloc(c, &c->d_ctw, (size_t)R * K * N));
                CHK(dalloc(c, &c->d_twnode, (size_t)R * K * PG_NODEG));
                CHK(dalloc(c, &c->d_joff, (size_t)R + 1));
                HIPCHK(c, hipHostMalloc((void**)&c->h_rad_p, (size_t)R * K * N * 4));
                HIPCHK(c, hipHostGetDevicePointer((void**)&c->hd_rad, c->h_rad_p, 0));
            }
            HIPCHK(c, hipMemcpyAsync(c->d_joff, c->h_joff.data(), ((size_t)R + 1) * 8, hipMemcpyHostToDevice, c->stream));
        }
    }
    // A kept graph stays lazy too when its reverse pass reads no node but the adopted ones (rows form, S <= 4096: pg_nodes_free
    // recomputes a node's row from its children; everything else that is read was somebody's child, i.e. adopted).
    const bool lazy_ok = !twist && (!graph || S <= 4096) && !(flags & PHYLO_EAGER_NODES) && !c->env.eager_nodes;
    // marks are plain stores and the extra launch costs less than the dead stores it removes at every size measured.
    // Sharded, the owner's write needs one more (tiny) collective per rank event (sweep_step_a); rehearsed with a
    // one-rank RCCL world (PHYLO_REHEARSE_SHARDED=1) the lazy sweep is 0.145 ms against 0.185 ms for the eager one at
    // primate.p's node size, more than a second collective costs
    const bool lazy = lazy_ok;
    int launches = 0;
    const bool fuse_scan = !twist && !graph && G == 1 && c->env.fuse_scan;   // opt-in: measured neutral alone, -4 % with 3 sweeps in flight
    c->swept = false;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    // one sweep alone on one GPU with lazy nodes: the adopted nodes are written in the bookkeeping launch (pk_rank_book_mat), found
    // by the resampling draws, which pk_sweep_prologue then leaves in d_rdraw.  Batched sweeps keep the separate launch (measured:
    // 2.62e11 against 2.64e11 units/s with the grouped form of the combined launch in round 2; round 3, a launch set alone: 3.57e11 against 3.77e11).
    const bool book_mat = lazy && c->world == 1 && c->comm.transport == 0 && !fuse_scan && N <= 64 && S <= 4096 && G == 1 && Kl <= 8192 &&
                          !c->env.separate_materialise && !c->env.book_one_per_wave;
    // sharded with lazy nodes: each owner finds ITS adopted nodes the same way (O(Kloc Kg / 64) comparisons) instead of every rank
    // searching the ancestors of all K particles (pk_all_marks, O(K) on every rank whatever the number of GPUs)
    const bool shard_form = c->world > 1 || (c->comm.transport != 0 && c->env.rehearse_sharded);
    const bool mat_by_draws = lazy && shard_form && !twist && !c->env.replicated_book && !c->env.separate_materialise && S <= 4096 &&
                              (((K / G) <= 4096 && Kl <= 8192) || ((K / G) % PK_MAT_GROUP == 0 && Kl % PK_MAT_GROUP == 0));
    const bool want_rdraw = book_mat || mat_by_draws;
    if (want_rdraw && !c->d_rdraw) CHK(dalloc(c, &c->d_rdraw, (size_t)R * K));
    const size_t mark_words = ((size_t)R * K + R + 3) & ~(size_t)3;
    int32_t* t_roots = graph ? c->d_hroots : c->d_roots[0];
    int32_t* t_cnt = graph ? c->d_hcnt : c->d_cnt[0];
    double* t_rootll = graph ? c->d_hrootll : c->d_rootll[0];
    if (!twist) {                                          // draws, initial tables and cleared marks: one launch
        if (G > 1) HIPCHK(c, hipMemcpyAsync(c->d_group_seeds, group_seeds, (size_t)G * 8, hipMemcpyHostToDevice, c->stream));
        pk_prologue_args pa{};
        pa.Q = c->d_Q; pa.lam_l = c->d_lam_l; pa.lam_r = c->d_lam_r; pa.jc = c->jc; pa.seed = seed; pa.R = R; pa.Kloc = Kl; pa.k0 = c->k0;
        pa.bl = c->d_bl; pa.br = c->d_br; pa.Pmat = c->d_Pmat; pa.Kg = K / G;
        pa.group_seeds = G > 1 ? (const uint64_t*)c->d_group_seeds : (const uint64_t*)nullptr;
        pa.rdraw = want_rdraw ? c->d_rdraw : (unsigned long long*)nullptr;
        pa.roots = t_roots; pa.cnt = t_cnt; pa.rootll = t_rootll; pa.nodell = c->d_nodell; pa.K = K; pa.N = N;
        pa.mark = lazy ? c->d_mark : (unsigned int*)nullptr;
        pa.mark_words = lazy ? (unsigned int)mark_words : 0u;
        // large launches (batched sweeps): the matrices sorted by Pade order inside workgroups of 1024 (pk_sweep_draws_sorted)
        const bool sorted = !c->jc && 2L * R * Kl >= 262144 && !c->env.no_sorted_draws;
        const int NT = sorted ? 256 : 64;
        pa.draw_blocks = sorted ? cdiv(2L * R * Kl, PK_DRAW_ITEMS) : cdiv(2L * R * Kl, 64);
        pa.init_blocks = cdiv((long)K * N, 4 * NT);
        pa.mark_blocks = lazy ? cdiv((long)mark_words, 4 * NT) : 0;
        const int rdraw_blocks = want_rdraw ? cdiv((long)(R - 1) * K, NT) : 0;
        const dim3 pgrid(pa.draw_blocks + pa.init_blocks + pa.mark_blocks + rdraw_blocks);
        if (sorted) hipLaunchKernelGGL(pk_sweep_prologue_sorted, pgrid, dim3(256), 0, c->stream, pa);
        else hipLaunchKernelGGL(pk_sweep_prologue, pgrid, dim3(64), 0, c->stream, pa);
        CHK(launch_check(c, "pk_sweep_prologue"));
        launches += 1;
    } else {
        if (lazy) HIPCHK(c, hipMemsetAsync(c->d_mark, 0, mark_words * sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(pk_init_tables, dim3(cdiv((long)K * N, 256)), dim3(256), 0, c->stream, t_roots, t_cnt, t_rootll,
                           (const double*)c->d_nodell, K, N);
        launches += 1;
    }
    CHK(launch_check(c, "pk_init_tables"));
    if (c->d_mirror) HIPCHK(c, hipMemsetAsync(c->d_mirror, 0, ((size_t)R * K + 4) * 4, c->stream));   // the cache of remote nodes is per sweep
    c->run = sweep_run{};
    c->run.seed = seed; c->run.flags = flags; c->run.M = M;
    c->run.twist = twist; c->run.graph = graph; c->run.lazy = lazy; c->run.timek = timek; c->run.fuse_scan = fuse_scan;
    c->run.book_mat = book_mat;
    c->run.mat_by_draws = mat_by_draws;
    c->run.launches = launches; c->run.next_r = 0; c->run.active = true;
    c->run.G = G;
    return PHYLO_OK;
}


// ---- the sweep as ONE launch (phylo_persist.h) ------------------------------------------------------------------
// Resident-workgroup kernels of different contexts must not be dispatched together: each waits inside the launch for ALL of
// its own workgroups, and two partially resident grids would wait for each other's CU slots for ever (the bounded spins turn
// that into a timeout error, not a hang).  So the one-launch sweeps of a process run one after the other on a device: every
// launch waits for the event the previous one recorded.  (Kernels of the launch path still overlap with them freely.)
static std::mutex g_persist_mu;
static hipEvent_t g_persist_done[64] = {};
static int persist_chain(phylo_ctx* c, bool after_launch) {
    std::lock_guard<std::mutex> lock(g_persist_mu);
    if (c->device < 0 || c->device >= 64) return fail(c, PHYLO_EINVAL, "device id out of range for the one-launch sweep");
    hipEvent_t& ev = g_persist_done[c->device];
    if (!after_launch) {
        if (ev) HIPCHK(c, hipStreamWaitEvent(c->stream, ev, 0));
        return PHYLO_OK;
    }
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(ev, c->stream));
    return PHYLO_OK;
}

// Plan: G groups x Wg resident workgroups, m = Kg / Wg particles each.  Returns false when this context / sweep is
// not eligible (the launch-per-rank-event path runs instead).
static bool persist_plan(phylo_ctx* c, uint32_t flags, int G, int* Wg_out, int* m_out) {
    if (!(c->env.one_launch || (flags & PHYLO_ONE_LAUNCH))) return false;             // opt-in
    if (c->world != 1 || c->comm.transport != 0) return false;                       // sharded: collectives between launches
    if (flags & (PHYLO_TWISTING | PHYLO_KEEP_GRAPH | PHYLO_EAGER_NODES | PHYLO_TIME_KERNELS)) return false;   // launch path only
    if (c->env.eager_nodes || c->env.fuse_scan || c->env.merge_pair_form || c->env.book_one_per_wave) return false;   // A/B switches of the launch path
    if (c->N > 32 || c->N < 2) return false;                                        // one wave per particle: a lane per root slot, history rows in lanes
    const int Kg = c->K / G;
    if (Kg > PP_MAX_KG) return false;                                               // the group's cdf lives in LDS
    // large nodes: the launch path spreads one node over several workgroups and its launches are long enough
    if ((double)c->S * 32.0 > 256.0 * 1024.0) return false;
    if (c->persist_blocks_per_cu < 0) {
        // residency of ONE workgroup per CU is all the kernel needs; ask the runtime whether it is admitted at all
        int nb = 0;
        const size_t lds = pp_layout(c->N, PP_CHUNK, 2048).total;
        hipError_t e = c->env.persist_nt == 512
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pp_sweep<512>, 512, lds)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pp_sweep<256>, 256, lds);
        c->persist_blocks_per_cu = (e == hipSuccess && nb >= 1) ? (nb >= 3 ? 2 : 1) : 0;   // one workgroup per CU by default; two only with a block of margin
    }
    if (c->persist_blocks_per_cu < 1 || c->n_cus < 1) return false;
    int Wt = c->env.persist_wgs > 0 ? c->env.persist_wgs : c->n_cus;                // default: one workgroup per CU
    if (Wt > c->n_cus * c->persist_blocks_per_cu) Wt = c->n_cus * c->persist_blocks_per_cu;
    if (Wt < G) return false;
    int Wg = Wt / G;
    if (Wg > Kg) Wg = Kg;
    while (Wg > 1 && Kg % Wg) --Wg;
    const int m = Kg / Wg;
    if (m > PP_MAX_M) return false;
    if (pp_layout(c->N, m, Kg).total > 150 * 1024) return false;
    *Wg_out = Wg;
    *m_out = m;
    return true;
}

static int sweep_persistent(phylo_ctx* c, uint64_t seed, uint32_t flags, const uint64_t* group_seeds, int G, int Wg, int m) {
    CHK(bind(c));
    c->run.active = false;
    if (!c->have_leaves || !c->have_model)
        return fail(c, PHYLO_ESTATE, "phylo_set_leaves and phylo_set_model must be called before a sweep");
    if (!c->state_ready) {
        CHK(ensure_sweep_state(c));
        CHK(refresh_leaf_ll(c));
    }
    const int N = c->N, K = c->K, S = c->S, R = N - 1, Kg = K / G;
    if (!c->d_rdraw) CHK(dalloc(c, &c->d_rdraw, (size_t)R * K));       // (the launch path's pk_rank_book_mat shares this buffer)
    if (!c->d_pctr) {
        CHK(dalloc(c, &c->d_pctr, (size_t)PK_MAX_GROUPS * PP_CTR_STRIDE));
        c->pctr_Wg = 0;
    }
    // Every group owns one monotone counter and expects it to equal ctr_base at launch.  Another grid shape, another number of
    // groups (the groups the previous launches did not use are behind) or a timed-out wait (partial arrivals): start again from 0.
    if (c->pctr_Wg != Wg || c->pctr_G != G || c->pctr_dirty) {
        HIPCHK(c, hipMemsetAsync(c->d_pctr, 0, (size_t)PK_MAX_GROUPS * PP_CTR_STRIDE * 8, c->stream));
        c->pctr_base = 0;
        c->pctr_Wg = Wg;
        c->pctr_G = G;
        c->pctr_dirty = false;
    }
    pp_args a{};
    a.N = N; a.S = S; a.K = K; a.Kg = Kg; a.G = G; a.R = R; a.Wg = Wg; a.m = m;
    a.T = c->site_tile;
    a.seed = seed; a.flags = flags; a.jc = c->jc;
    if (G > 1) {
        HIPCHK(c, hipMemcpyAsync(c->d_group_seeds, group_seeds, (size_t)G * 8, hipMemcpyHostToDevice, c->stream));
        a.group_seeds = c->d_group_seeds;
    }
    a.Q = c->d_Q; a.lam_l = c->d_lam_l; a.lam_r = c->d_lam_r; a.pi = c->d_pi; a.ldf = c->d_ldf;
    a.leaves = c->d_leaves; a.leaf_codes = c->leaves_coded ? c->d_leaf_codes : nullptr; a.pool = c->d_pool;
    for (int i = 0; i < 2; ++i) { a.roots[i] = c->d_roots[i]; a.cnt[i] = c->d_cnt[i]; a.rootll[i] = c->d_rootll[i]; }
    a.nodell = c->d_nodell; a.bl = c->d_bl; a.br = c->d_br; a.Pmat = c->d_Pmat; a.logw = c->d_logw; a.ll = c->d_ll;
    a.child = c->d_child; a.merges = c->d_merges; a.anc = c->d_anc; a.mark = c->d_mark;
    a.rdraw = c->d_rdraw;
    a.lse = c->d_lse; a.lse_stride = R + 1;
    a.ctr = c->d_pctr; a.ctr_base = c->pctr_base;
    a.timeout_word = c->d_counter + 1;
    if (c->env.persist_stamps) {
        if (!c->d_stamps) CHK(dalloc(c, &c->d_stamps, (size_t)(R + 1) * PP_NSTAMP));
        a.stamps = c->d_stamps;
    }
    const size_t lds = pp_layout(N, m, Kg).total;
    c->swept = false;
    CHK(persist_chain(c, false));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->env.persist_nt == 512) hipLaunchKernelGGL((pp_sweep<512>), dim3(G * Wg), dim3(512), lds, c->stream, a);
    else hipLaunchKernelGGL((pp_sweep<256>), dim3(G * Wg), dim3(256), lds, c->stream, a);
    CHK(launch_check(c, "pp_sweep"));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    CHK(persist_chain(c, true));
    c->pctr_base += (unsigned long long)R * Wg;
    c->swept = true;
    c->last_lazy = true;                                   // only adopted nodes are in the pool (marks say which)
    c->last_graph = false;
    c->last_G = G;
    c->last_final_missing = false;
    c->last_persistent = true;
    c->n_merge_events = 0;
    c->stats.n_launches = 1;
    c->stats.units = (double)K * S * R;
    c->stats.alg_bytes = 96.0 * c->stats.units;
    return PHYLO_OK;
}


int phylo_sweep_begin(phylo_ctx* c, uint64_t seed, uint32_t flags, int M) { return sweep_begin_impl(c, seed, flags, M, nullptr, 1); }

// All-gather of this rank's segments of `n_arrays` arrays (`count` doubles per rank; arrays[i] = the array's base, rank p's segment
// at + p * count), or with n_arrays == 0 a barrier across the ranks, on the context's stream.  With the exchange slab
// (c->p2p) it is ONE launch of pk_p2p_exchange -- peers' slabs written over xGMI, flags, bounded wait: no collective call, no
// second stream; otherwise the RCCL / host-mediated collective of phylo_comm.h.  purpose 0: the rank event's K-vectors; 1: barriers
// and the twisted proposal's choices (own flags and epochs: every rank issues the same sequence of each).
static int comm_exchange(phylo_ctx* c, double* const* arrays, int n_arrays, size_t count, int purpose) {
    if (c->comm.transport == 0) return PHYLO_OK;
    if (!c->p2p) {
        if (n_arrays == 0) { double* rows[1] = {c->d_sync}; return phylo_comm_allgather_inplace(c->comm, rows, 1, 1, c->stream, &c->err); }
        return phylo_comm_allgather_inplace(c->comm, arrays, n_arrays, count, c->stream, &c->err);
    }
    if (n_arrays > 4) return fail(c, PHYLO_EINVAL, "comm_exchange: at most four arrays");
    pk_p2p_args a{};
    a.slabs = c->d_xslab_ptrs; a.world = c->world; a.me = c->rank;
    a.n_seg = n_arrays; a.seg_count = (int)count;
    for (int i = 0; i < n_arrays; ++i) {
        const ptrdiff_t off = (const char*)arrays[i] - c->d_xslab;
        if (off < 0 || (size_t)off + (size_t)c->world * count * 8 > c->xslab_bytes) return fail(c, PHYLO_EINVAL, "comm_exchange: array outside the exchange slab");
        a.seg_off[i] = (size_t)off;
    }
    a.flag_off = c->x_flag_off[purpose];
    a.epoch = ++c->x_epoch[purpose];
    a.timeout_word = c->d_counter + 1;
    a.wait_ticks = c->env.p2p_wait_ticks;
    const size_t words = (size_t)n_arrays * count * (size_t)(c->world - 1);
    if (words > c->env.p2p_copy_words) {                   // large exchange: the copy over many workgroups, then the flags alone
        const size_t wgs = (words + 4095) / 4096;
        hipLaunchKernelGGL(pk_p2p_copy, dim3((unsigned)(wgs < 512 ? wgs : 512)), dim3(1024), 0, c->stream, a);
        CHK(launch_check(c, "pk_p2p_copy"));
        a.n_seg = 0;
    }
    hipLaunchKernelGGL(pk_p2p_exchange, dim3(1), dim3(1024), 0, c->stream, a);
    return launch_check(c, "pk_p2p_exchange");
}

// Sharded lazy nodes, first half of a rank event: every rank marks the nodes adopted at this resampling (the search
// of all K particles, no tables), the owner writes its marked nodes, and one tiny collective orders those writes before
// every rank's merge.  A no-op otherwise.  phylo_sweep_step runs it when the caller has not (phylo_sweep_step_a).
static int sweep_step_a(phylo_ctx* c) {
    CHK(bind(c));
    if (!c->run.active) return fail(c, PHYLO_ESTATE, "phylo_sweep_step without phylo_sweep_begin");
    const int N = c->N, K = c->K, Kl = c->Kloc, S = c->S, R = N - 1, r = c->run.next_r;
    if (r >= R) return fail(c, PHYLO_ESTATE, "all %d rank events of this sweep have been issued", R);
    c->run.a_done_r = r;
    const bool shard_form = c->world > 1 || (c->comm.transport != 0 && c->env.rehearse_sharded);   // the env: one-rank rehearsal
    if (!(shard_form && c->run.lazy && !c->run.twist && !c->env.replicated_book) || r == 0) return PHYLO_OK;
    const int G = c->run.G;
    pk_rank_args b{};
    b.r = r; b.n = N - r; b.N = N; b.S = S; b.K = K; b.Kloc = Kl; b.k0 = c->k0;
    b.seed = c->run.seed; b.flags = c->run.flags;
    b.cdf = c->d_cdf[r & 1];
    b.Kg = K / G; b.group_seeds = G > 1 ? c->d_group_seeds : nullptr;
    b.leaves = c->d_leaves; b.pool = c->d_pool; b.pool_ptrs = c->d_pool_ptrs;
    b.mirror = (c->run.twist || c->env.replicated_book) ? nullptr : c->d_mirror; b.cache = c->d_cache; b.cache_cap = c->cache_cap;
    b.leaf_codes = c->leaves_coded ? c->d_leaf_codes : nullptr;
    b.lazy = 1; b.mark = c->d_mark; b.child_all = c->d_child; b.Pmat_all = c->d_Pmat;
    if (c->run.mat_by_draws) {
        b.rdraw = c->d_rdraw + (size_t)r * K;
        const int grouped = ((K / G) > 4096 || Kl > 8192) ? 1 : 0;     // one workgroup per 64 particles when there are many
        hipLaunchKernelGGL(pk_materialize_by_draws, dim3(grouped ? Kl / PK_MAT_GROUP : Kl), dim3(PK_COLS), 0, c->stream, b, grouped);
        CHK(launch_check(c, "pk_materialize_by_draws"));
        c->run.launches += 1;
    } else {
        hipLaunchKernelGGL(pk_all_marks, dim3(cdiv(K, 4)), dim3(64), 0, c->stream, b);
        CHK(launch_check(c, "pk_all_marks"));
        // large launches of small nodes (batched sweeps): dispatching one workgroup per particle costs more than the few writes
        if (S <= 4096 && Kl > 8192) hipLaunchKernelGGL(pk_materialize_adopted_grouped, dim3(cdiv(Kl, PK_MAT_GROUP)), dim3(PK_COLS), 0, c->stream, b);
        else hipLaunchKernelGGL(pk_materialize_adopted, dim3(S <= 4096 ? 1 : cdiv(S, PK_MAT_TILE), Kl), dim3(PK_COLS), 0, c->stream, b);
        CHK(launch_check(c, "pk_materialize_adopted"));
        c->run.launches += 2;
    }
    CHK(comm_exchange(c, nullptr, 0, 0, 1));               // barrier: the owners' writes before every rank's merge
    return PHYLO_OK;
}

// phase 0: the whole rank event; 1: up to and including the merge; 2: what follows the all-gather of the rank
// event's three K-vectors (phylo_sweep_step_group issues that collective once for several sweeps)
static int sweep_step_impl(phylo_ctx* c, int phase) {
    CHK(bind(c));
    if (!c->run.active) return fail(c, PHYLO_ESTATE, "phylo_sweep_step without phylo_sweep_begin");
    const int N = c->N, K = c->K, Kl = c->Kloc, S = c->S, R = N - 1;
    if (c->run.next_r >= R) return fail(c, PHYLO_ESTATE, "all %d rank events of this sweep have been issued", R);
    const uint64_t seed = c->run.seed;
    const uint32_t flags = c->run.flags;
    const int M = c->run.M;
    const bool twist = c->run.twist, graph = c->run.graph, lazy = c->run.lazy, timek = c->run.timek, fuse_scan = c->run.fuse_scan;
    int launches = 0;
    const size_t lds = pk_book_lds_bytes(N);
    const size_t plane = (size_t)K * N;
    const int r = c->run.next_r;
    const int G = c->run.G, Kg = K / G;
    const double ll_tilde0 = pm_log(1.0 / (double)Kg);     // vcsmc.py:422
    const int cur = r & 1, nxt = cur ^ 1;
    if (phase != 2 && c->run.a_done_r != r) CHK(sweep_step_a(c));
    if (phase != 2) {
        pk_rank_args b{};
        b.r = r; b.n = N - r; b.N = N; b.S = S; b.K = K; b.Kloc = Kl; b.k0 = c->k0;
        b.seed = seed; b.flags = flags;
        b.Kg = Kg; b.group_seeds = G > 1 ? c->d_group_seeds : nullptr;
        // the node of the LAST rank event is never merged again: its log-likelihood is all the sweep needs
        b.no_store = (r == R - 1 && !graph && !(flags & PHYLO_EAGER_NODES) && !c->env.eager_nodes) ? 1 : 0;
        if (r == R - 1) c->run.final_missing = b.no_store && !lazy;
        if (graph) {                                       // every rank event keeps its tables: plane r -> plane r + 1
            b.roots_old = c->d_hroots + plane * r; b.cnt_old = c->d_hcnt + plane * r;
            b.roots_new = c->d_hroots + plane * (r + 1); b.cnt_new = c->d_hcnt + plane * (r + 1);
            b.rootll_old = c->d_hrootll + plane * r; b.rootll_new = c->d_hrootll + plane * (r + 1);
            b.pos_hist = c->d_pos + plane * r;
        } else {
            b.roots_old = c->d_roots[cur]; b.cnt_old = c->d_cnt[cur];
            b.roots_new = c->d_roots[nxt]; b.cnt_new = c->d_cnt[nxt];
            b.rootll_old = c->d_rootll[cur]; b.rootll_new = c->d_rootll[nxt];
        }
        b.cdf = c->d_cdf[cur];
        b.ll_prev = r > 0 ? c->d_ll + (size_t)(r - 1) * K : nullptr;
        b.nodell = c->d_nodell;
        b.ldf = c->d_ldf; b.ldf_n = N;
        b.bl = c->d_bl; b.br = c->d_br;
        b.lam_l = c->h_lam_l[r]; b.lam_r = c->h_lam_r[r];
        b.loglam_l = pm_log(b.lam_l); b.loglam_r = pm_log(b.lam_r);
        b.ll_tilde0 = ll_tilde0;
        b.leaves = c->d_leaves; b.pool = c->d_pool; b.pool_ptrs = c->d_pool_ptrs;
        // the cache of remote nodes is filled by the bookkeeping launch, which must come behind the owners' writes of this rank
        // event's adopted nodes: not so with replicated bookkeeping (and the twisted proposal reads its roots elsewhere)
        b.mirror = (twist || c->env.replicated_book) ? nullptr : c->d_mirror; b.cache = c->d_cache; b.cache_cap = c->cache_cap;
        b.leaf_codes = c->leaves_coded ? c->d_leaf_codes : nullptr;
        b.Pmat = c->d_Pmat + (size_t)r * Kl * 32;
        b.pi = c->d_pi;
        b.logw_r = c->d_logw + (size_t)r * K;
        b.ll_r = c->d_ll + (size_t)r * K;
        b.merges = c->d_merges; b.ancestors = c->d_anc;
        b.child = c->d_child + (size_t)r * Kl * 2; b.aux = c->d_aux;
        b.lazy = lazy ? 1 : 0; b.mark = c->d_mark; b.child_all = c->d_child; b.Pmat_all = c->d_Pmat;
        b.T = c->site_tile; b.ntiles = c->ntiles; b.tilev = c->d_tilev;
        if (twist) {
            pk_twist_args ta{};
            ta.a = b;
            ta.M = M;
            ta.J = ((N - r) * (N - r - 1) / 2) * M;
            ta.roots_ad = c->d_roots_ad; ta.cnt_ad = c->d_cnt_ad; ta.rootll_ad = c->d_rootll_ad;
            ta.tw_b = c->d_tw_b; ta.tw_P = c->d_tw_P; ta.pot = c->d_pot; ta.chosen = c->d_chosen;
            if (graph) {                                   // the reverse pass reads every rank event's sub-samples
                const size_t j0 = (size_t)c->h_joff[r];
                ta.tw_b = c->d_htw_b + j0 * 2; ta.tw_P = c->d_htw_P + j0 * 32; ta.pot = c->d_hpot + j0;
                ta.roots_ad = c->d_hroots_ad + plane * r; ta.chosen = c->d_hchosen + (size_t)r * K;
            }
            ta.Pmat_r = c->d_Pmat + (size_t)r * Kl * 32;
            ta.pair_hist = c->codes_valid ? c->d_pair_hist : nullptr;
            ta.codes = c->codes_valid ? c->d_leaf_codes : nullptr;
            ta.bl_r = c->d_bl + (size_t)r * Kl; ta.br_r = c->d_br + (size_t)r * Kl;
            ta.own_tables = c->comm.transport == 0 ? 1 : 0;
            ta.wbuf = c->d_twbuf;
            hipLaunchKernelGGL(pk_twist_adopt_draws, dim3(K + cdiv(2L * Kl * ta.J, 64)), dim3(64), 0, c->stream, ta, (const double*)c->d_Q, c->jc);
            CHK(launch_check(c, "pk_twist_adopt_draws"));
            if (ta.pair_hist) {                            // coded leaf-leaf pairs: 25 code pairs per row instead of S sites
                hipLaunchKernelGGL(pk_twist_potentials_ll, dim3((ta.J + 7) / 8, Kl), dim3(256), 0, c->stream, ta);
                CHK(launch_check(c, "pk_twist_potentials_ll"));
                ++launches;
            }
            // every other row: one wave each (at rank event 0 of a coded alignment every root is a leaf: nothing is left)
            const bool any_rows = !(ta.pair_hist && r == 0);
            const dim3 pgrid((unsigned)((((size_t)Kl * ta.J + 7) / 8) * 8));
            if (timek) {   // a twisted sweep's dominant kernel is this one: PHYLO_TIME_KERNELS stamps it instead of the merge
                if (any_rows) hipExtLaunchKernelGGL(pk_twist_potentials, pgrid, dim3(64), 0, c->stream, c->kev[2 * r], c->kev[2 * r + 1], 0, ta);
                else { HIPCHK(c, hipEventRecord(c->kev[2 * r], c->stream)); HIPCHK(c, hipEventRecord(c->kev[2 * r + 1], c->stream)); }
            } else if (any_rows) {
                hipLaunchKernelGGL(pk_twist_potentials, pgrid, dim3(64), 0, c->stream, ta);
            }
            CHK(launch_check(c, "pk_twist_potentials"));
            hipLaunchKernelGGL(pk_twist_choose, dim3(Kl), dim3(64), (size_t)(ta.J <= PK_TWIST_LDS_J ? ta.J : 0) * 8, c->stream, ta);
            CHK(launch_check(c, "pk_twist_choose"));
            if (c->comm.transport != 0) {
                double* rows[1] = {c->d_chosen};
                CHK(comm_exchange(c, rows, 1, (size_t)Kl, 1));
            }
            if (!ta.own_tables) {
                hipLaunchKernelGGL(pk_twist_tables, dim3(cdiv(K, 128)), dim3(128), 0, c->stream, ta);
                CHK(launch_check(c, "pk_twist_tables"));
                ++launches;
            }
            launches += 3;                                 // adopt + draws, potentials, choose
        } else if (c->run.book_mat && r > 0) {
            b.rdraw = c->d_rdraw + (size_t)r * K;
            const int mat_blocks = K;
            if (N <= 16) {
                const int bb = cdiv(K, PK_COLS / 16);
                hipLaunchKernelGGL(pk_rank_book_mat<16>, dim3(bb + mat_blocks), dim3(PK_COLS), lds * (PK_COLS / 16), c->stream, b, bb);
            } else if (N <= 32) {
                const int bb = cdiv(K, PK_COLS / 32);
                hipLaunchKernelGGL(pk_rank_book_mat<32>, dim3(bb + mat_blocks), dim3(PK_COLS), lds * (PK_COLS / 32), c->stream, b, bb);
            } else {                                       // 33..64 taxa (DS3-DS8): one wave per particle, four per workgroup
                const int bb = cdiv(K, PK_COLS / 64);
                hipLaunchKernelGGL(pk_rank_book_mat<64>, dim3(bb + mat_blocks), dim3(PK_COLS), lds * (PK_COLS / 64), c->stream, b, bb);
            }
            CHK(launch_check(c, "pk_rank_book_mat"));
            ++launches;
        } else if (r > 0 && fuse_scan) {
            // scan of log w_{r-1} and the bookkeeping of rank event r in one launch
            b.scan_logw = c->d_logw + (size_t)(r - 1) * K;
            b.scan_cdf = c->d_cdf[cur];
            b.scan_lse = c->d_lse + (r - 1);
            b.flag = c->d_counter; b.epoch = ++c->epoch; b.timeout_word = c->d_counter + 1;
            if (c->epoch == 0xffffffffu) c->epoch = 0;
            const size_t lds2 = lds > pk_scan_lds_bytes(K) ? lds : pk_scan_lds_bytes(K);
            hipLaunchKernelGGL(pk_rank_scan_book, dim3(K + 1), dim3(PK_COLS), lds2, c->stream, b);
            CHK(launch_check(c, "pk_rank_scan_book"));
            ++launches;
        } else {
            // sharded, plain proposal: every rank advances only ITS particles' root tables and reads an
            // adopted ancestor's row from the owner's slab over the peer mapping (ordered by the all-gather of the
            // previous rank event, like the node pool) instead of replicating the bookkeeping of all K particles
            const bool local_book = (c->world > 1 || (c->comm.transport != 0 && c->env.rehearse_sharded)) && !c->env.replicated_book;
            if (local_book) {
                b.tab_ptrs = c->d_tab_ptrs;
                b.tab_off_rootll = (size_t)cur * K * N * 8;
                b.tab_off_roots = (size_t)16 * K * N + (size_t)cur * K * N * 4;
                b.tab_off_cnt = (size_t)24 * K * N + (size_t)cur * K * N * 4;
            }
            c->run.local_book = local_book;
            const int nbook = local_book ? Kl : K;
            // large launches (batched sweeps) are bound by instruction issue: 8 lanes per particle serve 8 particles with one
            // instruction stream (3.52e11 -> 3.68e11 units/s for a launch set of 20 sweeps; 4 lanes: no further gain); small
            // launches are latency chains and keep the shorter 16-lane form
            if (N <= 16 && !c->env.book_one_per_wave && !c->env.book_lp16 && nbook >= 8192)
                hipLaunchKernelGGL(pk_rank_book_packed<8>, dim3(cdiv(nbook, 8)), dim3(64), lds * 8, c->stream, b);
            else if (N <= 16 && !c->env.book_one_per_wave)   // 4 particles per wave (PK_AUX + 2 = 10 <= 16 lanes)
                hipLaunchKernelGGL(pk_rank_book_packed<16>, dim3(cdiv(nbook, 4)), dim3(64), lds * 4, c->stream, b);
            else if (N <= 32 && !c->env.book_one_per_wave)   // 2 particles per wave
                hipLaunchKernelGGL(pk_rank_book_packed<32>, dim3(cdiv(nbook, 2)), dim3(64), lds * 2, c->stream, b);
            else
                hipLaunchKernelGGL(pk_rank_book, dim3(nbook), dim3(64), lds, c->stream, b);
            CHK(launch_check(c, "pk_rank_book"));
            ++launches;
        }
        if (lazy && r > 0 && !(c->run.local_book && !twist) && !c->run.book_mat) {   // (sharded with owner-held tables: done in sweep_step_a)
            // few nodes are marked, almost every workgroup leaves at once: one workgroup per particle for small nodes (a quarter
            // of the empty workgroups), site tiles for large ones (a marked node is then not limited to one CU's bandwidth)
            // large launches of small nodes (batched sweeps): dispatching one workgroup per particle costs more than the few writes
            if (S <= 4096 && Kl > 8192) hipLaunchKernelGGL(pk_materialize_adopted_grouped, dim3(cdiv(Kl, PK_MAT_GROUP)), dim3(PK_COLS), 0, c->stream, b);
            else hipLaunchKernelGGL(pk_materialize_adopted, dim3(S <= 4096 ? 1 : cdiv(S, PK_MAT_TILE), Kl), dim3(PK_COLS), 0, c->stream, b);
            CHK(launch_check(c, "pk_materialize_adopted"));
            ++launches;
            if (c->comm.transport != 0) CHK(comm_exchange(c, nullptr, 0, 0, 1));   // peers read these nodes in place: order them before every rank's merge
        }
        const bool nostore = (b.lazy || b.no_store) && !c->env.merge_pair_form;   // row-per-lane form when nothing is stored
        const size_t mitems = (size_t)Kl * c->ntiles;                              // one wave per (particle, site tile)
        const dim3 mgrid((unsigned)mitems);
        if (timek && !twist) {  // events stamped with the kernel's own begin/end (what rocprofv3 --kernel-trace reports)
            if (nostore) hipExtLaunchKernelGGL(pk_rank_merge_nostore, mgrid, dim3(64), 0, c->stream, c->kev[2 * r], c->kev[2 * r + 1], 0, b);
            else hipExtLaunchKernelGGL(pk_rank_merge, mgrid, dim3(PK_COLS), (size_t)(c->site_tile < S ? c->site_tile : S) * 8 + 16 + PK_STORE_STAGE_BYTES, c->stream, c->kev[2 * r], c->kev[2 * r + 1], 0, b);
        } else if (nostore) {
            hipLaunchKernelGGL(pk_rank_merge_nostore, mgrid, dim3(64), 0, c->stream, b);
        } else {
            hipLaunchKernelGGL(pk_rank_merge, mgrid, dim3(PK_COLS), (size_t)(c->site_tile < S ? c->site_tile : S) * 8 + 16 + PK_STORE_STAGE_BYTES, c->stream, b);
        }
        if (c->ntiles > 1) {    // rows longer than one tile: tile values left to right, then the particle's weight terms
            CHK(launch_check(c, "pk_rank_merge"));
            hipLaunchKernelGGL(pk_tile_epilogue, dim3(cdiv(Kl, 256)), dim3(256), 0, c->stream, b);
            ++launches;
        }
        CHK(launch_check(c, "pk_rank_merge"));
        ++launches;
    }
    if (phase == 0 && c->comm.transport != 0) {
        double* rows[3] = {c->d_logw + (size_t)r * K, c->d_ll + (size_t)r * K, c->d_nodell + N + (size_t)r * K};
        CHK(comm_exchange(c, rows, 3, (size_t)Kl, 0));
    }
    if (phase == 1) {
        c->run.launches += launches;
        return PHYLO_OK;
    }
    {
        {
            if (c->comm.transport != 0) {
                if (!c->run.local_book) {
                    hipLaunchKernelGGL(pk_fix_rootll, dim3(cdiv(K, 256)), dim3(256), 0, c->stream, c->d_rootll[nxt],
                                       (const double*)(c->d_nodell + N + (size_t)r * K), K, N, N - r, c->k0, Kl);
                    CHK(launch_check(c, "pk_fix_rootll"));
                    ++launches;
                }
            }
            if (G > 1) {                                       // one scan workgroup per batched sweep
                const bool fold = r + 1 == R && Kg <= PP_SCAN_KERNEL_MAX_KG;   // the last scan also sums the log-normalisers
                CHK(launch_scan(c, (const double*)(c->d_logw + (size_t)r * K), Kg, G, (r + 1 < R) ? c->d_cdf[nxt] : (uint64_t*)nullptr,
                                c->d_lse + r, R + 1, fold ? R : 0));
                if (fold) c->run.logz_done = true;
                ++launches;
            } else if (!fuse_scan || twist || r + 1 == R) {    // otherwise the next rank event's launch scans these weights
                const bool fold = r + 1 == R && K <= PP_SCAN_KERNEL_MAX_KG;
                CHK(launch_scan(c, (const double*)(c->d_logw + (size_t)r * K), K, 1, (r + 1 < R) ? c->d_cdf[nxt] : (uint64_t*)nullptr,
                                c->d_lse + r, 0, fold ? R : 0));
                if (fold) c->run.logz_done = true;
                ++launches;
            }
        }
    }
    c->run.launches += launches;
    ++c->run.next_r;
    return PHYLO_OK;
}

int phylo_sweep_step(phylo_ctx* c) { return sweep_step_impl(c, 0); }

int phylo_sweep_step_a(phylo_ctx* c) { return sweep_step_a(c); }

int phylo_sweep_step_group(phylo_ctx** ctxs, int n) {
    if (!ctxs || n < 1) return fail(nullptr, PHYLO_EINVAL, "phylo_sweep_step_group needs at least one context");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return fail(nullptr, PHYLO_EINVAL, "NULL context");
    if (n == 1 || ctxs[0]->comm.transport == 0) {          // nothing to fuse
        for (int i = 0; i < n; ++i) CHK(sweep_step_impl(ctxs[i], 0));
        return PHYLO_OK;
    }
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]->run.active || ctxs[i]->run.next_r != ctxs[0]->run.next_r)
            return fail(ctxs[i], PHYLO_ESTATE, "the sweeps of a group must be at the same rank event");
        if (&phylo_comm_link(ctxs[i]->comm) != &phylo_comm_link(ctxs[0]->comm))
            return fail(ctxs[i], PHYLO_EINVAL, "the contexts of a group must share one communicator (phylo_comm_share)");
    }
    for (int i = 0; i < n; ++i) CHK(sweep_step_impl(ctxs[i], 1));
    std::vector<phylo_comm*> comms(n);
    std::vector<double*> rows((size_t)3 * n);
    std::vector<hipStream_t> streams(n);
    for (int i = 0; i < n; ++i) {
        phylo_ctx* c = ctxs[i];
        const size_t r = (size_t)c->run.next_r, K = c->K;
        comms[i] = &c->comm;
        streams[i] = c->stream;
        rows[3 * i] = c->d_logw + r * K;
        rows[3 * i + 1] = c->d_ll + r * K;
        rows[3 * i + 2] = c->d_nodell + c->N + r * K;
    }
    if (ctxs[0]->p2p) {                                    // device-side exchange: every context has its own slab and flags
        for (int i = 0; i < n; ++i) {
            if (!ctxs[i]->p2p) return fail(ctxs[i], PHYLO_EINVAL, "the contexts of a group must use the same exchange");
            CHK(comm_exchange(ctxs[i], rows.data() + 3 * (size_t)i, 3, (size_t)ctxs[i]->Kloc, 0));
        }
    } else {
        phylo_ctx* c0 = ctxs[0];
        std::vector<size_t> counts(n);
        for (int i = 0; i < n; ++i) counts[i] = (size_t)ctxs[i]->Kloc;
        int rc = phylo_comm_allgather_group(comms.data(), rows.data(), 3, counts.data(), streams.data(), n, &c0->err);
        if (rc != PHYLO_OK) { g_last_error = c0->err; return rc; }
    }
    for (int i = 0; i < n; ++i) CHK(sweep_step_impl(ctxs[i], 2));
    return PHYLO_OK;
}

int phylo_sweep_finish(phylo_ctx* c) {
    CHK(bind(c));
    const int N = c->N, Kl = c->Kloc, S = c->S, R = N - 1;
    if (!c->run.active || c->run.next_r != R)
        return fail(c, PHYLO_ESTATE, "phylo_sweep_finish needs phylo_sweep_begin and all %d phylo_sweep_step calls", R);
    const uint32_t flags = c->run.flags;
    const int M = c->run.M;
    const bool twist = c->run.twist, graph = c->run.graph, lazy = c->run.lazy, timek = c->run.timek;
    int launches = c->run.launches;
    (void)flags;
    if (!c->run.logz_done) {
        if (c->run.G > 1)
            hipLaunchKernelGGL(pk_logz_total_groups, dim3(c->run.G), dim3(64), 0, c->stream, c->d_lse, R, R + 1);
        else
            hipLaunchKernelGGL(pk_logz_total, dim3(1), dim3(64), 0, c->stream, (const double*)c->d_lse, R, c->d_lse + R);
        CHK(launch_check(c, "pk_logz_total"));
        ++launches;
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    if (graph) {                                           // the reverse pass builds its lists from these on the host
        pg_copy3 cp{};                                     // (by a kernel into the pinned buffers: pg_copy_words says why)
        cp.src[0] = (const uint32_t*)c->d_anc; cp.dst[0] = c->hd_anc; cp.n[0] = R > 1 ? (size_t)(R - 1) * c->K * 2 : 0;
        cp.src[1] = (const uint32_t*)c->d_child; cp.dst[1] = c->hd_child; cp.n[1] = (size_t)R * c->K * 2;
        if (twist) { cp.src[2] = (const uint32_t*)c->d_hroots_ad; cp.dst[2] = c->hd_rad; cp.n[2] = (size_t)R * c->K * N; }
        // ... and what phylo_sweep_fetch would otherwise copy one by one: log Z-hat, the timeout word of the bounded waits
        cp.src[3] = (const uint32_t*)(c->d_lse + R); cp.dst[3] = c->hd_pub; cp.n[3] = 2;
        cp.src[4] = (const uint32_t*)(c->d_counter + 1); cp.dst[4] = c->hd_pub + 2; cp.n[4] = 1;
        const size_t words = cp.n[0] + cp.n[1] + cp.n[2];
        hipLaunchKernelGGL(pg_copy_words, dim3((unsigned)(words / 1024 < 1 ? 1 : (words / 1024 > 1024 ? 1024 : words / 1024))), dim3(256), 0, c->stream, cp);
        CHK(launch_check(c, "pg_copy_words"));
        HIPCHK(c, hipEventRecord(c->ev_gcopy, c->stream));
    }
    c->swept = true;
    c->run.active = false;
    c->last_lazy = lazy;
    c->last_graph = graph;
    c->last_graph_twist = graph && twist;
    c->last_graph_marks = graph && c->run.lazy;
    c->last_M = c->run.M;
    c->last_G = c->run.G;
    c->last_final_missing = c->run.final_missing;
    c->n_merge_events = timek ? R : 0;
    c->stats.n_launches = launches;
    c->stats.units = (double)Kl * S * R;
    c->stats.alg_bytes = 96.0 * c->stats.units;
    if (twist) {                                           // + K M S C(N+1,3) look-ahead merges, 64 B each (no store)
        const double ut = (double)Kl * M * S * ((double)(N + 1) * N * (N - 1) / 6.0);
        c->stats.units += ut;
        c->stats.alg_bytes += 64.0 * ut;
    }
    return PHYLO_OK;
}

int phylo_sweep_async(phylo_ctx* c, uint64_t seed, uint32_t flags, int M) {
    if (!c) return fail(nullptr, PHYLO_EINVAL, "ctx is NULL");
    int Wg = 0, m = 0;
    if (persist_plan(c, flags, 1, &Wg, &m)) return sweep_persistent(c, seed, flags, nullptr, 1, Wg, m);
    c->last_persistent = false;
    CHK(phylo_sweep_begin(c, seed, flags, M));
    for (int r = 0; r < c->N - 1; ++r) CHK(phylo_sweep_step(c));
    return phylo_sweep_finish(c);
}

int phylo_sweep_batch_begin(phylo_ctx* c, const uint64_t* seeds, int G, uint32_t flags) {
    if (!c) return fail(nullptr, PHYLO_EINVAL, "ctx is NULL");
    if (!seeds) return fail(c, PHYLO_EINVAL, "seeds is NULL");
    c->h_group_seeds.assign(seeds, seeds + (G > 0 ? G : 0));          // stays alive until the copy has run
    return sweep_begin_impl(c, G > 0 ? seeds[0] : 0, flags, 1, c->h_group_seeds.data(), G);
}

int phylo_sweep_batch_async(phylo_ctx* c, const uint64_t* seeds, int G, uint32_t flags) {
    if (!c) return fail(nullptr, PHYLO_EINVAL, "ctx is NULL");
    int Wg = 0, m = 0;
    if (seeds && G >= 1 && G <= PK_MAX_GROUPS && c->K % G == 0 && persist_plan(c, flags, G, &Wg, &m)) {
        c->h_group_seeds.assign(seeds, seeds + G);          // stays alive until the copy has run
        return sweep_persistent(c, seeds[0], flags, c->h_group_seeds.data(), G, Wg, m);
    }
    c->last_persistent = false;
    CHK(phylo_sweep_batch_begin(c, seeds, G, flags));
    for (int r = 0; r < c->N - 1; ++r) CHK(phylo_sweep_step(c));
    return phylo_sweep_finish(c);
}

static int check_timeout_word(phylo_ctx* c, bool pub);

int phylo_sweep_fetch_logz(phylo_ctx* c, double* logZ, int G) {
    CHK(bind(c));
    if (!c->swept) return fail(c, PHYLO_ESTATE, "no sweep has been run");
    if (!logZ || G != c->last_G) return fail(c, PHYLO_EINVAL, "the last sweep batched %d sweep(s), asked for %d", c->last_G, G);
    const size_t R = (size_t)c->N - 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CHK(check_timeout_word(c, false));
    std::vector<double> all((R + 1) * G);
    HIPCHK(c, hipMemcpy(all.data(), c->d_lse, all.size() * 8, hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g) logZ[g] = all[g * (R + 1) + R];
    return PHYLO_OK;
}

int phylo_synchronize(phylo_ctx* c) {
    CHK(bind(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

// The word every bounded wait between workgroups sets when it gives up (one-launch sweep, fused scan + bookkeeping): a sweep that
// timed out is invalid.  The stream must be idle.  `pub`: the sweep's copy kernel left the word in pinned memory.
static int check_timeout_word(phylo_ctx* c, bool pub) {
    unsigned int tmo = 0;
    if (pub) tmo = c->h_pub[2];
    else HIPCHK(c, hipMemcpy(&tmo, c->d_counter + 1, sizeof tmo, hipMemcpyDeviceToHost));
    if (tmo) {
        HIPCHK(c, hipMemset(c->d_counter + 1, 0, sizeof tmo));
        c->pctr_dirty = true;                              // the one-launch sweep's arrival counters hold partial arrivals
        return fail(c, PHYLO_EHIP, "a bounded wait between workgroups timed out inside a launch; results are invalid");
    }
    return PHYLO_OK;
}

int phylo_sweep_fetch(phylo_ctx* c, double* log_weights, double* log_lik, double* lbranch, double* rbranch,
                      int32_t* merges, int64_t* ancestors, double* logZ, phylo_stats* perf) {
    CHK(bind(c));
    if (!c->swept) return fail(c, PHYLO_ESTATE, "no sweep has been run");
    const size_t R = (size_t)c->N - 1, K = c->K, Kl = c->Kloc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const bool pub = c->last_graph && c->h_pub;            // the sweep kept its graph: its copy kernel left these in pinned memory
    CHK(check_timeout_word(c, pub));
    // log_weights / log_lik are stored with global columns; hand back this rank's columns
    if (Kl == K) {                      // one rank: rows are contiguous
        if (log_weights) HIPCHK(c, hipMemcpy(log_weights, c->d_logw, R * K * 8, hipMemcpyDeviceToHost));
        if (log_lik) HIPCHK(c, hipMemcpy(log_lik, c->d_ll, R * K * 8, hipMemcpyDeviceToHost));
    } else {
        if (log_weights)
            HIPCHK(c, hipMemcpy2D(log_weights, Kl * 8, c->d_logw + c->k0, K * 8, Kl * 8, R, hipMemcpyDeviceToHost));
        if (log_lik) HIPCHK(c, hipMemcpy2D(log_lik, Kl * 8, c->d_ll + c->k0, K * 8, Kl * 8, R, hipMemcpyDeviceToHost));
    }
    if (lbranch) HIPCHK(c, hipMemcpy(lbranch, c->d_bl, R * Kl * 8, hipMemcpyDeviceToHost));
    if (rbranch) HIPCHK(c, hipMemcpy(rbranch, c->d_br, R * Kl * 8, hipMemcpyDeviceToHost));
    if (merges) HIPCHK(c, hipMemcpy(merges, c->d_merges, R * Kl * 2 * 4, hipMemcpyDeviceToHost));
    if (ancestors && R > 1) HIPCHK(c, hipMemcpy(ancestors, c->d_anc, (R - 1) * Kl * 8, hipMemcpyDeviceToHost));
    if (logZ) {
        if (pub) memcpy(logZ, c->h_pub, 8);
        else HIPCHK(c, hipMemcpy(logZ, c->d_lse + R, 8, hipMemcpyDeviceToHost));
    }
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->stats.sweep_ms = ms;
    c->stats.merge_ms = 0.0;
    c->stats.merge_launches = c->n_merge_events;
    for (int r = 0; r < c->n_merge_events; ++r) {
        float km = 0.f;
        HIPCHK(c, hipEventElapsedTime(&km, c->kev[2 * r], c->kev[2 * r + 1]));
        c->stats.merge_ms += km;
    }
    if (perf) *perf = c->stats;
    return PHYLO_OK;
}

int phylo_sweep(phylo_ctx* c, uint64_t seed, uint32_t flags, int M, double* log_weights, double* log_lik,
                double* lbranch, double* rbranch, int32_t* merges, int64_t* ancestors, double* logZ,
                phylo_stats* perf) {
    CHK(phylo_sweep_async(c, seed, flags, M));
    return phylo_sweep_fetch(c, log_weights, log_lik, lbranch, rbranch, merges, ancestors, logZ, perf);
}

int phylo_sweep_node(phylo_ctx* c, int r, int k, double* out) {
    CHK(bind(c));
    if (!c->swept) return fail(c, PHYLO_ESTATE, "no sweep has been run");
    if (r < 0 || r >= c->N - 1 || k < 0 || k >= c->Kloc || !out) return fail(c, PHYLO_EINVAL, "bad (r, k)");
    if (c->last_lazy) {                 // write every node that the lazy sweep skipped, oldest rank event first
        pk_rank_args b{};
        b.N = c->N; b.S = c->S; b.K = c->K; b.Kloc = c->Kloc; b.k0 = c->k0;
        b.leaves = c->d_leaves; b.pool = c->d_pool; b.pool_ptrs = c->d_pool_ptrs;
        b.mark = c->d_mark; b.child_all = c->d_child; b.Pmat_all = c->d_Pmat;
        for (int rho = 0; rho < c->N - 1; ++rho) {       // sharded: a collective (every rank must call phylo_sweep_node)
            hipLaunchKernelGGL(pk_materialize_rank, dim3(c->Kloc), dim3(PK_COLS), 0, c->stream, b, rho);
            CHK(launch_check(c, "pk_materialize_rank"));
            c->last_graph_marks = false;                   // the marks now cover more than the adopted nodes
            if (c->comm.transport != 0) {
                CHK(comm_exchange(c, nullptr, 0, 0, 1));
            }
        }
        c->last_lazy = false;
    }
    if (c->last_final_missing && r == c->N - 2) {        // the sweep did not store the last rank event's nodes: write them now
        pk_rank_args b{};
        b.N = c->N; b.S = c->S; b.K = c->K; b.Kloc = c->Kloc; b.k0 = c->k0;
        b.leaves = c->d_leaves; b.pool = c->d_pool; b.pool_ptrs = c->d_pool_ptrs;
        b.child_all = c->d_child; b.Pmat_all = c->d_Pmat;
        hipLaunchKernelGGL(pk_materialize_all, dim3(c->Kloc), dim3(PK_COLS), 0, c->stream, b, c->N - 2);
        CHK(launch_check(c, "pk_materialize_all"));
        c->last_final_missing = false;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t node_sz = (size_t)c->S * 4;
    HIPCHK(c, hipMemcpy(out, c->d_pool + ((size_t)r * c->Kloc + k) * node_sz, node_sz * 8, hipMemcpyDeviceToHost));
    return PHYLO_OK;
}

// ---- the reverse pass's integer lists built on the device (phylo_revlists_dev.h): launches, then the few integers the host needs
struct dl_meta {
    std::vector<int32_t> ev_adp0, ev_slow0;
    int32_t n_adp = 0, n_chunks = 0, n_slow = 0, n_par = 0;
};
static unsigned bit_length(size_t v) { unsigned b = 0; while (v) { ++b; v >>= 1; } return b ? b : 1; }
extern "C++" {
template <int ITEMS>
static int dev_lists_adopters(phylo_ctx* c, const pg_dl_args& d, hipStream_t s) {
    const size_t lds = pg_dl_sort<ITEMS>::storage_bytes + (size_t)d.K * 4;
    static bool raised = false;                            // (beyond the 64 KB every kernel may ask for: say so once)
    if (lds > 65536 && !raised) {
        HIPCHK(c, hipFuncSetAttribute((const void*)pg_dl_adopters<ITEMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    hipLaunchKernelGGL(pg_dl_adopters<ITEMS>, dim3(d.R), dim3(PG_DL_BLOCK), lds, s, d, bit_length((size_t)d.K));
    return launch_check(c, "pg_dl_adopters");
}
}  // extern "C++"
// The list kernels on sL; ev_dl is recorded when the lists that the coefficient chain and the host need are there (pg_dl_lists);
// the parents' sort on sS (sort = false: the caller issues it later with dev_lists_sort).  Nothing here waits for the host;
// dev_lists_wait does.
static int dev_lists_launch(phylo_ctx* c, hipStream_t sL, hipStream_t sS, bool kernels = true, bool sort = true) {
    const int N = c->N, K = c->K, R = N - 1;
    const size_t nn = (size_t)R * K, nb = (nn + PG_DL_BLOCK - 1) / PG_DL_BLOCK;
    const size_t meta_ints = (size_t)PG_DL_META_INTS(R);
    void* ws = nullptr;
    CHK(scratch_get(c, 8, (8 * nn + 4 * nb + meta_ints + 32) * 4, &ws));
    pg_dl_args d{};
    d.N = N; d.R = R; d.K = K;
    d.anc = c->d_anc; d.child = c->d_child;
    int32_t* w = (int32_t*)ws;
    d.cnt_par = w; d.ticket = w + nn; w += nn + 16;
    d.adopted = w; w += nn;
    d.bsum = w; w += 4 * nb;
    d.dmeta = w; w += meta_ints;
    d.pkey = (uint32_t*)w; d.pval = (uint32_t*)w + 2 * nn; w += 4 * nn;
    uint32_t* pkey_out = (uint32_t*)w;
    d.L = pg_lists_carve(c->d_ad_off, (size_t)R, (size_t)K);
    d.meta = c->hd_dlmeta;
    const unsigned pbits = bit_length(2 * nn);
    if (c->dl_temp_nn != nn) {                             // (the size query launches nothing)
        size_t tp = 0;
        HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tp, d.pkey, pkey_out, d.pval, (uint32_t*)d.L.par_idx, 2 * nn, 0u, pbits, sL));
        c->dl_temp_p = tp; c->dl_temp_nn = nn;
    }
    void* tp = nullptr;
    CHK(scratch_get(c, 9, c->dl_temp_p + 16, &tp));
    if (kernels) {
        if (K <= 1024) CHK(dev_lists_adopters<1>(c, d, sL));
        else if (K <= 2048) CHK(dev_lists_adopters<2>(c, d, sL));
        else if (K <= 4096) CHK(dev_lists_adopters<4>(c, d, sL));
        else CHK(dev_lists_adopters<8>(c, d, sL));
        if (R > 1) {
            hipLaunchKernelGGL(pg_dl_count, dim3(cdiv((long)(2 * nn - 2 * (size_t)K), 256)), dim3(256), 0, sL, d);
            CHK(launch_check(c, "pg_dl_count"));
        }
        hipLaunchKernelGGL(pg_dl_sums, dim3((unsigned)nb), dim3(PG_DL_BLOCK), 0, sL, d);
        CHK(launch_check(c, "pg_dl_sums"));
        hipLaunchKernelGGL(pg_dl_lists, dim3((unsigned)nb), dim3(PG_DL_BLOCK), 0, sL, d);
        CHK(launch_check(c, "pg_dl_lists"));
        HIPCHK(c, hipEventRecord(c->ev_dl, sL));
    }
    if (!sort) return PHYLO_OK;
    sL = sS;
    // the parents' sort: six small launches through rocPRIM's host code, 8 us of host time each -- captured once per shape (the
    // buffers are the context's own and stay where they are), replayed with one call
    const void* key[4] = {ws, tp, (const void*)c->d_ad_off, (const void*)nn};
    if (c->dl_graph && memcmp(key, c->dl_graph_key, sizeof key) != 0) {
        (void)hipGraphExecDestroy(c->dl_graph);
        c->dl_graph = nullptr;
    }
    if (!c->dl_graph && !c->dl_no_graph) {
        hipGraph_t gr = nullptr;
        bool ok = hipStreamBeginCapture(sL, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            size_t bytes = c->dl_temp_p;
            const hipError_t e1 = rocprim::radix_sort_pairs(tp, bytes, d.pkey, pkey_out, d.pval, (uint32_t*)d.L.par_idx, 2 * nn, 0u, pbits, sL);
            const hipError_t e2 = hipStreamEndCapture(sL, &gr);
            ok = e1 == hipSuccess && e2 == hipSuccess && gr && hipGraphInstantiate(&c->dl_graph, gr, nullptr, nullptr, 0) == hipSuccess;
            if (gr) (void)hipGraphDestroy(gr);
        }
        if (!ok) {                                         // no capture on this runtime: the plain launches every time
            (void)hipGetLastError();
            c->dl_graph = nullptr;
            c->dl_no_graph = true;
        } else {
            memcpy(c->dl_graph_key, key, sizeof key);
        }
    }
    if (c->dl_graph) {
        HIPCHK(c, hipGraphLaunch(c->dl_graph, sL));
        } else {
        size_t bytes = c->dl_temp_p;
        HIPCHK(c, rocprim::radix_sort_pairs(tp, bytes, d.pkey, pkey_out, d.pval, (uint32_t*)d.L.par_idx, 2 * nn, 0u, pbits, sL));
    }
    return PHYLO_OK;
}
static int dev_lists_wait(phylo_ctx* c, dl_meta& m) {
    CHK(wait_event_spin(c, c->ev_dl));
    const int R = c->N - 1;
    const int32_t* h = c->h_dlmeta;
    m.ev_adp0.assign(h, h + R + 1);
    m.ev_slow0.assign(h + R + 1, h + 2 * (R + 1));
    const int32_t* t = h + 2 * (R + 1);
    m.n_adp = t[0]; m.n_chunks = t[1]; m.n_slow = t[2]; m.n_par = t[3];
    return PHYLO_OK;
}

static int sweep_backward_impl(phylo_ctx* c, double* d_lam_l, double* d_lam_r, double* d_pi, double* d_Q, phylo_stats* perf);
// reverse passes in flight in this process (several host threads, each with its own context): a launch that waits inside the GPU for
// another launch of its own pass (pg_nodes_rows_all beside pg_coeff_all) assumes the two share the GPU with nobody who waits likewise
static std::atomic<int> g_backward_in_flight{0};

int phylo_sweep_backward(phylo_ctx* c, double* d_lam_l, double* d_lam_r, double* d_pi, double* d_Q, phylo_stats* perf) {
    struct in_flight { in_flight() { ++g_backward_in_flight; } ~in_flight() { --g_backward_in_flight; } } guard;
    const int rc = sweep_backward_impl(c, d_lam_l, d_lam_r, d_pi, d_Q, perf);
    if (rc != PHYLO_OK && c) {
        // an early return may have left kernels on the side streams that still read the pinned list image and the graph: join them
        // before anything rebuilds or frees those, and drop the graph (the next backward needs a new sweep)
        if (c->gstream) (void)hipStreamSynchronize(c->gstream);
        if (c->bgstream) (void)hipStreamSynchronize(c->bgstream);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->last_graph = false;
    }
    return rc;
}

static int sweep_backward_impl(phylo_ctx* c, double* d_lam_l, double* d_lam_r, double* d_pi, double* d_Q, phylo_stats* perf) {
    CHK(bind(c));
    if (!c->swept || !c->last_graph)
        return fail(c, PHYLO_ESTATE, "phylo_sweep_backward needs a preceding sweep with PHYLO_KEEP_GRAPH");
    const int N = c->N, K = c->K, S = c->S, R = N - 1;
    const bool rows_form = S <= 4096;                     // pg_nodes_rows: one workgroup per node, one tile
    const int T = rows_form ? 1 : (S + PG_NT - 1) / PG_NT;
    const bool twist = c->last_graph_twist;
    const size_t nn = (size_t)R * K;
    // ---- what does not need the integer lists is launched first: the GPU works while the host builds them
    pg_args g{};
    g.N = N; g.S = S; g.K = K; g.R = R; g.T = T; g.jc = c->jc;
    g.twist = twist ? 1 : 0;
    g.leaves = c->d_leaves; g.pool = c->d_pool; g.adj = c->d_adj; g.Pmat = c->d_Pmat;
    g.bl = c->d_bl; g.br = c->d_br; g.logw = c->d_logw; g.lse = c->d_lse;
    g.pi = c->d_pi; g.Q = c->d_Q; g.lam_l = c->d_lam_l; g.lam_r = c->d_lam_r;
    g.child = c->d_child; g.pos = c->d_pos; g.roots = c->d_hroots;
    g.ad_off = c->d_ad_off; g.ad_idx = c->d_ad_idx; g.par_off = c->d_par_off; g.par_idx = c->d_par_idx;
    g.heavy_first = c->d_heavy; g.chunk_beg = c->d_chunk_beg; g.chunk_cnt = c->d_chunk_cnt;
    g.slow_flag = c->d_slow_flag; g.slow_idx = c->d_slow_idx;
    g.om = c->d_om; g.G = c->d_G; g.C = c->d_C; g.part = c->d_part; g.nodeg = c->d_nodeg;
    g.leafpi = c->d_leafpi; g.leafterm = c->d_leafterm; g.terms = c->d_terms; g.out = c->d_gout;
    if (twist) {
        g.tw.M = c->last_M; g.tw.joff = c->d_joff; g.tw.roots_ad = c->d_hroots_ad;
        g.tw.tw_b = c->d_htw_b; g.tw.tw_P = c->d_htw_P; g.tw.pot = c->d_hpot; g.tw.chosen = c->d_hchosen;
        g.tw.tau = c->d_tau; g.tw.ctw = c->d_ctw; g.tw.twpart = c->d_twpart; g.tw.twnode = c->d_twnode;
        g.tw.pair_hist = (c->codes_valid && c->hist_ready) ? c->d_pair_hist : nullptr;
        const size_t J0 = (size_t)((N * (N - 1)) / 2) * c->last_M;
        void* sl = nullptr;
        CHK(scratch_get(c, 4, (size_t)K * ((J0 + 255) / 256) * PG_NODEG * 8, &sl));
        g.tw.twslice = (double*)sl;
    }
    HIPCHK(c, hipEventRecord(c->evb0, c->stream));
    const int nrk = cdiv((long)R * K, 256);
    // a lazy sweep left marks: a node nobody adopted has no parents and alpha = omega, known without any list -- nearly all
    // nodes, done while the lists are built
    const bool early_free = rows_form && !twist && c->last_graph_marks;
    // After a lazy sweep with the plain proposal the lists are built by kernels (phylo_revlists_dev.h) and the host waits for a few
    // dozen integers; PHYLO_REV_HOST_LISTS keeps the host builders (the A/B switch, and what every other form uses).
    const bool dev_lists = early_free && !c->env.rev_host_lists && c->Kloc == K && K <= PG_DL_MAX_K;
    // The list kernels need the sweep's ancestors and children and nothing else: they are queued right behind the sweep on its own
    // stream, ahead of the early kernels below (they head the longest chain: lists -> sort -> chunk sums -> adopted nodes).
    // (their sort goes to the second stream as soon as the host has seen the sweep end: queued there behind the lists' event, it
    //  neither waits for the host to read the counts nor holds up the coefficient chain on this stream)
    const bool sort_early = dev_lists && !c->env.grad_sort_late && !c->env.grad_one_stream;
    if (dev_lists) CHK(dev_lists_launch(c, c->stream, c->stream, true, false));
    hipLaunchKernelGGL(pg_omega, dim3(nrk), dim3(256), 0, c->stream, g);
    CHK(launch_check(c, "pg_omega"));
    hipLaunchKernelGGL(pg_leafpi, dim3(N), dim3(256), 0, c->stream, g);
    CHK(launch_check(c, "pg_leafpi"));
    g.alpha_om = early_free ? 1 : 0;                       // a free parent then is a node nobody adopted: alpha = omega
    // That launch is 85 us of throughput work nothing waits for before pg_node_finish, while everything else below is a chain of
    // small dependent launches: it runs on a stream of the lowest priority, in the background of the chains.
    // (Measured, K = 2048: reverse pass 0.539 -> 0.511 ms with all 898 sites; with 256 sites the launch is 25 us and the extra
    //  events and the fill launch cost more than they hide, 0.440 -> 0.473 ms: large sweeps only.)
    const bool bg_free = early_free && !c->env.grad_one_stream && (c->env.grad_two_streams || nn * (size_t)S >= ((size_t)12 << 20));
    if (early_free) {
        g.mark = c->d_mark;
        if (bg_free) {
            hipLaunchKernelGGL(pg_fill_free, dim3(nrk), dim3(256), 0, c->stream, g);
            CHK(launch_check(c, "pg_fill_free"));
            HIPCHK(c, hipEventRecord(c->ev_bgfork, c->stream));
            // (its launch follows the wait for the sweep's end below: a second queue with a pending wait slows the sweep's own
            //  dependent launches by half a microsecond each -- 19 us per sweep, measured)
        } else {
            hipLaunchKernelGGL(pg_nodes_free, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, c->stream, g, 0);
            CHK(launch_check(c, "pg_nodes_free"));
        }
    }
    int tw_launches = 0;
    if (twist) {
        const size_t J0 = (size_t)((N * (N - 1)) / 2) * c->last_M;
        hipLaunchKernelGGL(pg_twist_tau, dim3(R * K), dim3(64), (J0 <= 8192 ? J0 : 8192) * 8, c->stream, g);   // later rank events have fewer rows and use LDS
        CHK(launch_check(c, "pg_twist_tau"));
        hipLaunchKernelGGL(pg_twist_pbar, dim3((unsigned)((c->h_joff[R] + 3) / 4)), dim3(256), 0, c->stream, g);
        CHK(launch_check(c, "pg_twist_pbar"));
        if (g.tw.pair_hist)
            for (int r = 0; r < R; ++r) {
                const long rows_r = (long)(c->h_joff[r + 1] - c->h_joff[r]);
                hipLaunchKernelGGL(pg_twist_pbar_ll, dim3(cdiv(rows_r, 64)), dim3(64), 0, c->stream, g, r);
                CHK(launch_check(c, "pg_twist_pbar_ll"));
                ++tw_launches;
            }
        for (int r = 0; r < R; ++r) {
            const int Jr = (((N - r) * (N - r - 1)) / 2) * c->last_M;
            const int KB = Jr >= 256 ? 1 : 256 / Jr;
            const int nsl = Jr > 256 ? cdiv(Jr, 256) : 1;
            hipLaunchKernelGGL(pg_twist_finish, dim3(cdiv(K, KB), nsl), dim3(256), 0, c->stream, g, r);
            CHK(launch_check(c, "pg_twist_finish"));
            ++tw_launches;
            if (nsl > 1) {
                hipLaunchKernelGGL(pg_twist_finish_sum, dim3(cdiv((long)K * PG_NODEG, 256)), dim3(256), 0, c->stream, g, r, nsl);
                CHK(launch_check(c, "pg_twist_finish_sum"));
                ++tw_launches;
            }
        }
        tw_launches += 2;
    }
    // ---- integer bookkeeping of the reverse pass: who adopted whom, and which nodes have which parents.  The sweep left the
    //      ancestors and children in pinned host memory (asynchronous copies behind its last launch); the lists are built straight
    //      into the pinned image of the device slab (ad_off | ad_idx | par_off | par_idx | heavy | chunk_beg | chunk_cnt).
    CHK(wait_event_spin(c, c->ev_gcopy));
    auto launch_bg_free = [&]() -> int {
        HIPCHK(c, hipStreamWaitEvent(c->bgstream, c->ev_bgfork, 0));
        hipLaunchKernelGGL(pg_nodes_free, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, c->bgstream, g, 3);
        CHK(launch_check(c, "pg_nodes_free"));
        HIPCHK(c, hipEventRecord(c->ev_bgdone, c->bgstream));
        return PHYLO_OK;
    };
    dl_meta dm;
    if (bg_free && !dev_lists) CHK(launch_bg_free());
    const auto host_t0 = std::chrono::steady_clock::now();
    const int64_t* anc = c->h_anc_p;
    const int32_t* child = c->h_child_p;
    const pg_lists L = pg_lists_carve(c->h_csr_p, (size_t)R, (size_t)K);     // (phylo_revlists.h: the builders, tested on the CPU)
    int32_t* const ad_off = L.ad_off;
    int32_t* const par_off = L.par_off;
    int32_t* const heavy = L.heavy;
    int32_t* const slow_flag = L.slow_flag;
    int32_t* const adp = L.adp;
    const size_t cap = L.cap;
    if (!dev_lists) pg_lists_clear(L, R, K);
    std::vector<int32_t>& cur = c->h_cur;
    // Two chains of small dependent launches remain, both newest rank event first: the coefficients (on the context's stream) and
    // the adopted nodes' adjoints, which need the coefficients of their own and of later rank events only (ev_coeff[r]), on a
    // second stream.  After the early pg_nodes_free the parents' lists are built FIRST (they need of the adopters only who was
    // adopted: pg_mark_adopted), so that the one launch over all heavy nodes' free parents runs while the host sorts the adopters
    // and beside the coefficient chain; the adopted nodes' chain then follows the coefficients one rank event behind.
    // (Without that reordering and for small sweeps two streams gain nothing -- the host finishes the parents' lists only when the
    //  coefficient chain is over -- and the events cost 13 us: primate.p, K = 2048, 0.542 against 0.555 ms; DS1, K = 4096: 2.48 -> 2.16.)
    // (Measured, K = 2048: reverse pass 0.522 -> 0.476 ms with all 898 sites, 0.455 -> 0.466 with 256: large sweeps only, like the
    //  background launch.)
    const bool reorder = bg_free;
    const bool two = !c->env.grad_one_stream && (c->env.grad_two_streams || nn >= 65536 || reorder || dev_lists);
    hipStream_t sB = two ? c->gstream : c->stream;
    // (the host has seen the sweep end: the second stream needs no event to start on its outputs, and the list kernels run
    //  beside the early kernels)
    if (dev_lists) {
        // (the list kernels run on the context's stream, ahead of the coefficient chain)
        if (sort_early) {
            HIPCHK(c, hipStreamWaitEvent(c->gstream, c->ev_dl, 0));
            CHK(dev_lists_launch(c, c->gstream, c->gstream, false, true));
        }
        // the list kernels are workgroups of 1024 threads that everything else waits for: on a GPU that the background launch has
        // filled they wait for a whole free CU each, kernel after kernel (lists ready after 120 us instead of 55): the background
        // launch starts behind them.  (Measured, primate.p K = 2048 / DS1 K = 4096, reverse pass: background launch first 0.504 /
        //  1.634 ms, behind the lists 0.486 / 1.653, behind the parents' sort 0.510 / 1.746; a high-priority second stream
        //  changes nothing.)
        if (bg_free) {
            HIPCHK(c, hipStreamWaitEvent(c->bgstream, c->ev_dl, 0));
            CHK(launch_bg_free());
        }
    }
    if (two) {
        HIPCHK(c, hipEventRecord(c->ev_gfork, c->stream));                 // everything launched so far (the early kernels)
        HIPCHK(c, hipStreamWaitEvent(sB, c->ev_gfork, 0));
    }
    std::vector<int32_t> rank_chunk0, ev_slow0;
    pg_parents_info pinfo{};
    size_t max_chunks = 0, n_chunks = 0;
    if (early_free && !dev_lists) pg_mark_adopted(R, K, anc, L);
    // the adopted nodes' chain as ONE launch (pg_nodes_rows_all; the plain proposal with the lists built on the device): the
    // coefficient chain -- then the longest chain of the pass -- is issued first and in one go, the parents' sort and the chunk sums
    // behind it, and the one launch waits for the last coefficients
    bool rows_all = false, rows_overlap = false;
    auto launch_chunks = [&]() -> int {
        // The parents of a heavy node are nearly all nodes nobody merged again: their share of the node's adjoint needs their
        // alpha = omega and nothing else.  ONE launch sums
        // them for the chunks of all rank events; the chain below is then pg_nodes_rows alone, which adds the flagged parents.
        const size_t rowlen = (size_t)S * 4;
        for (size_t cbeg = 0; cbeg < n_chunks; cbeg += 65535) {
            const size_t cn = n_chunks - cbeg < 65535 ? n_chunks - cbeg : 65535;
            pg_args g2 = g;
            g2.cpart = g.cpart + cbeg * rowlen;
            if (c->env.grad_quad_chunks) hipLaunchKernelGGL(pg_parent_chunks, dim3(cdiv(S, 16 * PG_CSTEPS), (unsigned)cn), dim3(256), 0, sB, g2, (int)cbeg);
            else hipLaunchKernelGGL(pg_parent_chunks_rows, dim3(cdiv(S, 64), (unsigned)cn), dim3(256), 0, sB, g2, (int)cbeg);
            CHK(launch_check(c, "pg_parent_chunks"));
        }
        return PHYLO_OK;
    };
    auto parents_block = [&]() -> int {
    // ---- parents, heavy nodes' chunks, flagged nodes by rank event (pg_build_parents, or what pg_dl_lists reports)
    if (dev_lists) {
        CHK(dev_lists_wait(c, dm));
        ev_slow0 = dm.ev_slow0;
        pinfo.n_chunks = (size_t)dm.n_chunks; pinfo.max_chunks = 0; pinfo.n_slow = dm.n_slow; pinfo.n_par = dm.n_par;
    } else {
        pinfo = pg_build_parents(N, R, K, child, rows_form, early_free, L, cur, rank_chunk0, ev_slow0);
    }
    max_chunks = pinfo.max_chunks; n_chunks = pinfo.n_chunks;
    {
        const int32_t ns = pinfo.n_slow;
        void* cpart = nullptr;
        // rows form: the chunk sums of ALL rank events are produced by one launch (free parents only: nothing of the chain is
        // needed for them), so the buffer holds every chunk; else one rank event's at a time
        CHK(scratch_get(c, 5, (early_free ? n_chunks : max_chunks) * (size_t)S * 4 * 8, &cpart));
        g.cpart = (double*)cpart;
        g.chunks_free_only = early_free ? 1 : 0;
        g.TS = cdiv(S, 256);
        {
            void* fp = nullptr;
            CHK(scratch_get(c, 2, ((nn + 31) / 32) * 20 * 8, &fp));
            g.fin_part = (double*)fp;
        }
        if (rows_form) {
            void* sp = nullptr;
            CHK(scratch_get(c, 3, (size_t)(ns ? ns : 1) * g.TS * PG_PART * 8, &sp));
            g.slowpart = (double*)sp;
        }
    }
    if (!dev_lists) {   // what was used of everything between the adopters' lists and the adopted particles, by a kernel (pg_copy_words)
        pg_copy3 cp{};
        const size_t o0 = (size_t)(par_off - ad_off), o1 = (size_t)(heavy - ad_off), o2 = (size_t)(slow_flag - ad_off);
        cp.src[0] = c->hd_csr + o0; cp.dst[0] = (uint32_t*)(c->d_ad_off + o0); cp.n[0] = nn + 1 + (size_t)par_off[nn];   // par_off | par_idx
        cp.src[1] = c->hd_csr + o1; cp.dst[1] = (uint32_t*)(c->d_ad_off + o1); cp.n[1] = nn + cap + n_chunks;            // heavy | chunk_beg | chunk_cnt
        cp.src[2] = c->hd_csr + o2; cp.dst[2] = (uint32_t*)(c->d_ad_off + o2); cp.n[2] = nn + (size_t)ev_slow0[R];       // slow_flag | slow_idx
        const size_t words = cp.n[0] + cp.n[1] + cp.n[2];
        hipLaunchKernelGGL(pg_copy_words, dim3((unsigned)(words / 1024 < 1 ? 1 : (words / 1024 > 1024 ? 1024 : words / 1024))), dim3(256), 0, sB, cp);
        CHK(launch_check(c, "pg_copy_words"));
        if (two) HIPCHK(c, hipEventRecord(c->ev_gup, sB));
    }
    rows_all = rows_form && early_free && dev_lists && two && !c->env.grad_rows_chain && pinfo.n_slow > 0 &&
               (size_t)pinfo.n_slow * (size_t)g.TS <= 16384;
    if (rows_all) {
        if (!c->d_row_done) {
            const size_t words = (size_t)R * K * (size_t)cdiv(c->S, 256) + 2 * (size_t)R;   // + coeff_done[R] | coeff_ticket[R]
            CHK(dalloc(c, &c->d_row_done, words));
            HIPCHK(c, hipMemsetAsync(c->d_row_done, 0, words * 4, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));     // (once per context: ahead of every launch, on any stream, that touches them)
            c->row_epoch = 0;
        }
        if (++c->row_epoch == 0) ++c->row_epoch;             // (0 is what the words hold before their first pass)
        g.row_done = c->d_row_done;
        g.row_epoch = c->row_epoch;
        g.row_timeout = (unsigned int*)(c->hd_dlmeta + PG_DL_META_INTS(R));
        // few enough workgroups to leave the coefficient chain room on every SIMD: the launch runs BESIDE that chain and waits, rank
        // event by rank event, for its completion words; else it is launched behind the chain's last event
        rows_overlap = (size_t)pinfo.n_slow * (size_t)g.TS <= 512 && !c->env.grad_rows_no_overlap && g_backward_in_flight.load() <= 1;
        g.coeff_done = c->d_row_done + (size_t)R * K * (size_t)cdiv(c->S, 256);
        g.coeff_ticket = g.coeff_done + R;
        g.coeff_mask = 0ull;
        for (int r = 0; r + 1 < R; ++r)
            if (dm.ev_adp0[r + 1] > dm.ev_adp0[r]) g.coeff_mask |= 1ull << r;
    }
    // (the launch behind the coefficient chain: that chain is the longer one and is issued first; the launch beside it: the sort and
    //  the chunk sums first, so that the adopted nodes follow the coefficients rank event by rank event)
    const bool q3_first = !rows_all || rows_overlap;
    if (dev_lists && q3_first && !sort_early) CHK(dev_lists_launch(c, sB, sB, false, true));   // the parents' sort (the host has seen the list kernels end)
    if (early_free && n_chunks > 0 && q3_first) CHK(launch_chunks());
        return PHYLO_OK;
    };
    if (reorder || dev_lists) CHK(parents_block());
    std::vector<int32_t> ev_adp0;                          // adopted particles of rank event r: adp[ev_adp0[r] .. ev_adp0[r + 1])
    int32_t n_adp = 0;
    if (dev_lists) { ev_adp0 = dm.ev_adp0; n_adp = dm.n_adp; }
    else n_adp = pg_build_adopters(R, K, anc, L, cur, ev_adp0);
    // the adopters' lists are all the coefficient chain needs: it runs while the host goes on with the parents' lists.  When the
    // early pg_nodes_free has dealt with everybody nobody adopted, the chain runs over the adopted particles alone.
    const size_t ad_ints = (size_t)R * (K + 1) + nn;
    // With the parents' lists already there, the two chains are launched in turn, a rank event of each: the host needs ~3 us per
    // call, and the adopted nodes' chain queued behind all the coefficient launches would start ~70 us late.
    const bool interleave = early_free && two && (reorder || dev_lists) && !rows_all;
    auto launch_coeff = [&](int r) -> int {
        const int na = ev_adp0[r + 1] - ev_adp0[r];
        if (na > 0) {
            hipLaunchKernelGGL(pg_coeff, dim3(na, cdiv(N - r - 1, 4)), dim3(256), 0, c->stream, g, r, (int)ev_adp0[r]);
            CHK(launch_check(c, "pg_coeff"));
        }
        if (two) HIPCHK(c, hipEventRecord(c->ev_coeff[r], c->stream));
        return PHYLO_OK;
    };
    if (!dev_lists) HIPCHK(c, hipMemcpyAsync(c->d_ad_off, c->h_csr_p, ad_ints * 4, hipMemcpyHostToDevice, c->stream));
    if (early_free) {
        if (!dev_lists) HIPCHK(c, hipMemcpyAsync(c->d_adp, adp, (size_t)(n_adp ? n_adp : 1) * 4, hipMemcpyHostToDevice, c->stream));
        g.adp = c->d_adp;
        if (n_adp > 0) {
            hipLaunchKernelGGL(pg_G, dim3(n_adp), dim3(64), 0, c->stream, g);
            CHK(launch_check(c, "pg_G"));
        }
        if (two) HIPCHK(c, hipEventRecord(c->ev_coeff[R - 1], c->stream));   // (the last rank event has no adopters: C is there)
        // with the adopted nodes in one launch, the coefficient chain is one launch too (pg_coeff_all) when all of its workgroups
        // can be resident
        long coeff_wgs = 0;
        for (int r = R - 2; r >= 0; --r) coeff_wgs += (long)(ev_adp0[r + 1] - ev_adp0[r]) * cdiv(N - r - 1, 4);
        if (rows_all && R - 1 <= 64 && coeff_wgs > 0 && coeff_wgs <= 2048 && !c->env.grad_coeff_chain) {
            pg_coeff_plan pl{};
            int at = 0;
            for (int r = R - 2; r >= 0; --r) {
                pl.first[r] = at; pl.adp0[r] = ev_adp0[r]; pl.ny[r] = cdiv(N - r - 1, 4);
                at += (ev_adp0[r + 1] - ev_adp0[r]) * pl.ny[r];
            }
            hipLaunchKernelGGL(pg_coeff_all, dim3((unsigned)at), dim3(256), 0, c->stream, g, pl);
            CHK(launch_check(c, "pg_coeff_all"));
            if (two) HIPCHK(c, hipEventRecord(c->ev_coeff[0], c->stream));
        } else if (!interleave) {
            for (int r = R - 2; r >= 0; --r) CHK(launch_coeff(r));
        }
    } else {
        hipLaunchKernelGGL(pg_G, dim3(R * K), dim3(64), 0, c->stream, g);
        CHK(launch_check(c, "pg_G"));
        for (int r = R - 1; r >= 0; --r) {
            hipLaunchKernelGGL(pg_coeff, dim3(K, cdiv(N - r - 1, 4)), dim3(256), 0, c->stream, g, r, 0);
            CHK(launch_check(c, "pg_coeff"));
            if (two) HIPCHK(c, hipEventRecord(c->ev_coeff[r], c->stream));
        }
    }
    if (!interleave) {
        hipLaunchKernelGGL(pg_leafterm, dim3(cdiv(K, 256)), dim3(256), 0, c->stream, g);
        CHK(launch_check(c, "pg_leafterm"));
    }
    // twisted proposal: the look-ahead merges of rank event r touch every internal node among the adopted roots.  Entries
    // (adopter, slot) grouped by node (ascending adopter), cut into chunks of PG_XCH; lists for all rank events in one upload.
    std::vector<int32_t> ev_chunk0((size_t)R + 1, 0), ev_node0((size_t)R + 1, 0);
    size_t tw_max_chunks = 0;
    void *d_xlists = nullptr, *d_tpart = nullptr;
    size_t n_xent = 0, n_xchunks = 0, n_xnodes = 0;
    if (twist) {
        const int32_t* rad = c->h_rad_p;                            // pinned copy made when the sweep ended
        std::vector<int32_t> xent, xc_node, xc_beg, xc_cnt, xc_part, xn_id, xn_c0, xn_nc;
        std::vector<int32_t> cnt, first;
        for (int r = 0; r < R; ++r) {
            ev_chunk0[r] = (int32_t)xc_node.size();
            ev_node0[r] = (int32_t)xn_id.size();
            if (r == 0) continue;                                   // rank event 0 adopts leaves only
            const int n = N - r;
            const size_t nn_r = (size_t)r * K;                      // nodes that exist before rank event r
            cnt.assign(nn_r + 1, 0);
            const int32_t* tab = rad + (size_t)r * K * N;
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < n; ++i) {
                    const int x = tab[(size_t)k * N + i];
                    if (x >= N) ++cnt[(size_t)(x - N) + 1];
                }
            for (size_t i = 0; i < nn_r; ++i) cnt[i + 1] += cnt[i];
            const size_t base = xent.size();
            xent.resize(base + (size_t)cnt[nn_r]);
            first.assign(cnt.begin(), cnt.end() - 1);
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < n; ++i) {
                    const int x = tab[(size_t)k * N + i];
                    if (x >= N) xent[base + (size_t)first[x - N]++] = k * N + i;
                }
            // chunk shape of this rank event: enough workgroups to fill the GPU, not more rows than pg_twist_xsum should add per node.
            // Many entries (large K): up to PG_XCH entries per chunk, all partner slots.  Few entries (the K = 32..64 of the
            // reference's experiments): one entry per chunk and the n - 1 partner slots cut into slices, or a chunk is one thread's
            // walk over (n - 1) M merges per site, a few hundred microseconds with M = 10.
            const long total_ent = cnt[nn_r];
            const long target = 2048 / cdiv(S, 256) > 64 ? 2048 / cdiv(S, 256) : 64;
            int xch = (int)((total_ent + target - 1) / target);
            xch = xch < 1 ? 1 : (xch > PG_XCH ? PG_XCH : xch);
            int slices = 1;
            if (xch == 1 && total_ent > 0) {
                slices = (int)(target / total_ent);
                slices = slices < 1 ? 1 : (slices > n ? n : slices);
            }
            const int pw = (n + slices - 1) / slices;              // partner slots per slice
            for (size_t x = 0; x < nn_r; ++x) {
                const int m = cnt[x + 1] - cnt[x];
                if (m == 0) continue;
                xn_id.push_back((int32_t)(x + N));
                xn_c0.push_back((int32_t)xc_node.size());
                int nc = 0;
                for (int b = 0; b < m; b += xch)
                    for (int p0 = 0; p0 < n; p0 += pw) {
                        xc_node.push_back((int32_t)(x + N));
                        xc_beg.push_back((int32_t)(base + cnt[x] + b));
                        xc_cnt.push_back(m - b < xch ? m - b : xch);
                        xc_part.push_back(p0 | ((p0 + pw < n ? p0 + pw : n) << 16));
                        ++nc;
                    }
                xn_nc.push_back(nc);
            }
            const size_t nch = xc_node.size() - (size_t)ev_chunk0[r];
            if (nch > tw_max_chunks) tw_max_chunks = nch;
        }
        ev_chunk0[R] = (int32_t)xc_node.size();
        ev_node0[R] = (int32_t)xn_id.size();
        n_xent = xent.size(); n_xchunks = xc_node.size(); n_xnodes = xn_id.size();
        // the newest rank event that touches a node is launched first: its pg_twist_xsum starts the node's adjoint row (bit 30)
        // instead of adding to it, so nothing has to be cleared; such a node goes through pg_nodes_rows (flag bit 1)
        for (size_t i = n_xnodes; i-- > 0;) {
            int32_t& f = slow_flag[xn_id[i] - N];
            if (!(f & 2)) { f |= 2; xn_nc[i] |= 1 << 30; }
        }
        std::vector<int32_t>& pk = c->h_xlists;
        pk.resize(n_xent + 4 * n_xchunks + 3 * n_xnodes + 1);
        int32_t* w = pk.data();
        auto put = [&](const std::vector<int32_t>& v) { if (!v.empty()) memcpy(w, v.data(), v.size() * 4); w += v.size(); };
        put(xent); put(xc_node); put(xc_beg); put(xc_cnt); put(xc_part); put(xn_id); put(xn_c0); put(xn_nc);
        CHK(scratch_get(c, 6, pk.size() * 4, &d_xlists));
        HIPCHK(c, hipMemcpyAsync(d_xlists, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, sB));
        CHK(scratch_get(c, 7, tw_max_chunks * (size_t)S * 4 * 8, &d_tpart));
    }
    if (!reorder && !dev_lists) CHK(parents_block());
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    if (twist) {
        const int32_t* xl = (const int32_t*)d_xlists;
        g.tw.xent = xl; xl += n_xent;
        g.tw.xchunk_node = xl; xl += n_xchunks;
        g.tw.xchunk_beg = xl; xl += n_xchunks;
        g.tw.xchunk_cnt = xl; xl += n_xchunks;
        g.tw.xchunk_part = xl; xl += n_xchunks;
        g.tw.xnode_id = xl; xl += n_xnodes;
        g.tw.xnode_chunk0 = xl; xl += n_xnodes;
        g.tw.xnode_nchunks = xl;
        g.tw.tpart = (double*)d_tpart;
    }
    int node_launches = early_free ? 1 : 0;
    if (rows_form && !early_free) {                        // the sweep left no marks: every node nobody merged again, now
        ++node_launches;
        if (two) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gup, 0));   // (needs the flags of the second upload)
        hipLaunchKernelGGL(pg_nodes_free, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, c->stream, g, 1);
        CHK(launch_check(c, "pg_nodes_free"));
    }
    if (rows_all && !rows_overlap) {                       // (behind the coefficient launches: they head the longer chain)
        if (!sort_early) CHK(dev_lists_launch(c, sB, sB, false, true));
        if (n_chunks > 0) CHK(launch_chunks());
    }
    if (rows_all) {
        if (!rows_overlap) HIPCHK(c, hipStreamWaitEvent(sB, c->ev_coeff[0], 0));   // every alpha is there
        hipLaunchKernelGGL(pg_nodes_rows_all, dim3((unsigned)pinfo.n_slow, g.TS), dim3(256), 0, sB, g, (int)pinfo.n_slow);
        CHK(launch_check(c, "pg_nodes_rows_all"));
        ++node_launches;
    }
    for (int r = rows_all ? -1 : R - 1; r >= 0; --r) {
        if (interleave && r >= 1) CHK(launch_coeff(r - 1));
        if (two) HIPCHK(c, hipStreamWaitEvent(sB, c->ev_coeff[r], 0));
        if (twist && ev_chunk0[r + 1] > ev_chunk0[r]) {
            hipLaunchKernelGGL(pg_twist_xchunks, dim3(ev_chunk0[r + 1] - ev_chunk0[r], cdiv(S, 256)), dim3(256), 0, sB, g, r, (int)ev_chunk0[r]);
            CHK(launch_check(c, "pg_twist_xchunks"));
            hipLaunchKernelGGL(pg_twist_xsum, dim3(ev_node0[r + 1] - ev_node0[r], cdiv((long)S * 4, 256)), dim3(256), 0, sB, g, (int)ev_node0[r], (int)ev_chunk0[r]);
            CHK(launch_check(c, "pg_twist_xsum"));
            tw_launches += 2;
        }
        const int nch = early_free ? 0 : rank_chunk0[r + 1] - rank_chunk0[r];   // (after the early pg_nodes_free: summed above, all rank events at once)
        if (nch > 0) {
            if (c->env.grad_quad_chunks) hipLaunchKernelGGL(pg_parent_chunks, dim3(cdiv(S, 16 * PG_CSTEPS), nch), dim3(256), 0, sB, g, (int)rank_chunk0[r]);
            else hipLaunchKernelGGL(pg_parent_chunks_rows, dim3(cdiv(S, 64), nch), dim3(256), 0, sB, g, (int)rank_chunk0[r]);
            CHK(launch_check(c, "pg_parent_chunks"));
            ++node_launches;
        }
        if (rows_form) {
            const int nslow = ev_slow0[r + 1] - ev_slow0[r];
            if (nslow > 0) {
                hipLaunchKernelGGL(pg_nodes_rows, dim3(nslow, g.TS), dim3(256), 0, sB, g, r, (int)ev_slow0[r],
                                   early_free && !dev_lists ? (int)rank_chunk0[r] : 0);   // (device-built lists: heavy[] is the global chunk index)
                ++node_launches;
            }
        } else {
            hipLaunchKernelGGL(pg_nodes, dim3(T, K), dim3(256), 0, sB, g, r);
            ++node_launches;
        }
        CHK(launch_check(c, "pg_nodes"));
    }
    if (interleave) {
        hipLaunchKernelGGL(pg_leafterm, dim3(cdiv(K, 256)), dim3(256), 0, c->stream, g);
        CHK(launch_check(c, "pg_leafterm"));
    }
    if (two) {
        HIPCHK(c, hipEventRecord(c->ev_gjoin, sB));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gjoin, 0));
    }
    if (bg_free) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_bgdone, 0));
    hipLaunchKernelGGL(pg_node_finish, dim3(cdiv((long)R * K, 32)), dim3(256), 0, c->stream, g);
    CHK(launch_check(c, "pg_node_finish"));
    hipLaunchKernelGGL(pg_scalars, dim3(nrk), dim3(256), 0, c->stream, g);
    CHK(launch_check(c, "pg_scalars"));
    hipLaunchKernelGGL(pg_reduce, dim3(2 * R + 20), dim3(256), 0, c->stream, g);
    CHK(launch_check(c, "pg_reduce"));
    HIPCHK(c, hipEventRecord(c->evb1, c->stream));
    std::vector<double> out((size_t)2 * R + 20);
    HIPCHK(c, hipMemcpyAsync(out.data(), c->d_gout, out.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_gjoin, c->stream));     // (free again: the stream has waited for it above)
    CHK(wait_event_spin(c, c->ev_gjoin));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (rows_all && c->h_dlmeta[PG_DL_META_INTS(R)] != 0) {
        c->h_dlmeta[PG_DL_META_INTS(R)] = 0;
        return fail(c, PHYLO_EHIP, "reverse pass: a workgroup of pg_nodes_rows_all gave up waiting for a parent's adjoint tile "
                                       "(PHYLO_GRAD_ROWS_CHAIN=1 runs a launch per rank event instead)");
    }
    if (d_lam_l) memcpy(d_lam_l, out.data(), (size_t)R * 8);
    if (d_lam_r) memcpy(d_lam_r, out.data() + R, (size_t)R * 8);
    if (d_pi) memcpy(d_pi, out.data() + 2 * R, 4 * 8);
    if (d_Q) memcpy(d_Q, out.data() + 2 * R + 4, 16 * 8);
    if (perf) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->evb0, c->evb1));
        *perf = c->stats;
        perf->sweep_ms = ms;
        perf->n_launches = R + 7 + node_launches + tw_launches;
        perf->merge_ms = host_ms;                          // here: host time of the integer lists (built, or waited for: device lists)
        perf->merge_launches = dev_lists ? 1 : 0;          // here: 1 = the lists were built by kernels (phylo_revlists_dev.h)
    }
    return PHYLO_OK;
}

// ---- a VI training step's host half in C++ (phylo_train.h): variables -> model -> sweep + reverse pass -> gradients of the variables
int phylo_vi_gradients(phylo_ctx* c, uint64_t seed, uint32_t flags, int M, int jc, const double* vars, double* logZ, double* grads,
                       phylo_stats* fwd, phylo_stats* bwd) {
    CHK(bind(c));
    if (!vars || !grads) return fail(c, PHYLO_EINVAL, "phylo_vi_gradients: NULL argument");
    const int R = c->N - 1;
    double Q[16], pi[4];
    std::vector<double>& lam = c->h_vi_lam;
    lam.resize((size_t)2 * R);
    for (int r = 0; r < 2 * R; ++r) lam[r] = std::exp(vars[r]);
    if (jc) pt_jc_Q(Q); else pt_get_Q(vars + 2 * R, Q);
    pt_get_pi(vars + 2 * R + 16, pi);
    CHK(phylo_set_model(c, Q, pi, lam.data(), lam.data() + R, jc));
    CHK(phylo_sweep_async(c, seed, flags | PHYLO_KEEP_GRAPH, M));
    double raw[2 * 64 + 20];
    if (R > 64) return fail(c, PHYLO_EINVAL, "phylo_vi_gradients: at most 65 taxa");
    CHK(phylo_sweep_backward(c, raw, raw + R, raw + 2 * R, raw + 2 * R + 4, bwd));
    CHK(phylo_sweep_fetch(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, logZ, fwd));
    pt_chain_rules(R, jc, Q, pi, lam.data(), lam.data() + R, raw, raw + R, raw + 2 * R, raw + 2 * R + 4, grads);
    return PHYLO_OK;
}

int phylo_vi_apply(int n_taxa, int jc, double* vars, const double* grads, int kind, double lr, double beta1, double beta2, double eps, int64_t* t,
                   double* m, double* v) {
    if (n_taxa < 2 || !vars || !grads || (kind != 0 && (!t || !m || !v)))
        return fail(nullptr, PHYLO_EINVAL, "phylo_vi_apply: bad arguments");
    const int R = n_taxa - 1;
    pt_apply(jc ? 2 * R : 2 * R + 20, vars, grads, kind, lr, beta1, beta2, eps, t, m, v);
    return PHYLO_OK;
}

int phylo_debug_stamps(phylo_ctx* c, uint64_t* out, int n) {
    CHK(bind(c));
    if (!c->d_stamps) return fail(c, PHYLO_ESTATE, "no stamps: create the context with PHYLO_PERSIST_STAMPS=1 and run a one-launch sweep");
    const int have = c->N * PP_NSTAMP;
    if (!out || n < have) return fail(c, PHYLO_EINVAL, "phylo_debug_stamps needs room for N * %d = %d values", PP_NSTAMP, have);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_stamps, (size_t)have * 8, hipMemcpyDeviceToHost));
    return PHYLO_OK;
}

int phylo_debug_reverse_lists(int N, int K, const int64_t* ancestors, const int32_t* child, int early_free, int rows_form,
                              const int32_t* lookahead_nodes, int n_lookahead, int32_t* lists, int64_t n_lists, int32_t* meta, int n_meta) {
    if (N < 2 || K < 1 || !child || !lists || !meta || (N > 2 && !ancestors) || (n_lookahead > 0 && !lookahead_nodes))
        return fail(nullptr, PHYLO_EINVAL, "phylo_debug_reverse_lists: bad arguments");
    const int R = N - 1;
    if (n_lists < (int64_t)pg_lists_ints((size_t)R, (size_t)K) || n_meta < 6 + 3 * (R + 1))
        return fail(nullptr, PHYLO_EINVAL, "phylo_debug_reverse_lists: lists needs %zu ints, meta %d", pg_lists_ints((size_t)R, (size_t)K), 6 + 3 * (R + 1));
    const pg_lists L = pg_lists_carve(lists, (size_t)R, (size_t)K);
    pg_lists_clear(L, R, K);
    std::vector<int32_t> cur, ev_adp0, rank_chunk0, ev_slow0;
    const int32_t n_adp = pg_build_adopters(R, K, ancestors, L, cur, ev_adp0);
    if (early_free) pg_mark_adopted(R, K, ancestors, L);
    for (int i = 0; i < n_lookahead; ++i) {
        if (lookahead_nodes[i] < N || lookahead_nodes[i] >= N + R * K) return fail(nullptr, PHYLO_EINVAL, "phylo_debug_reverse_lists: node id out of range");
        L.slow_flag[lookahead_nodes[i] - N] |= 2;
    }
    const pg_parents_info o = pg_build_parents(N, R, K, child, rows_form != 0, rows_form != 0 && early_free != 0, L, cur, rank_chunk0, ev_slow0);
    meta[0] = n_adp; meta[1] = (int32_t)o.n_chunks; meta[2] = (int32_t)o.max_chunks; meta[3] = o.n_slow; meta[4] = o.n_par;
    meta[5] = (int32_t)L.cap;
    for (int r = 0; r <= R; ++r) {
        meta[6 + r] = ev_adp0[r];
        meta[6 + (R + 1) + r] = rank_chunk0[r];
        meta[6 + 2 * (R + 1) + r] = ev_slow0[r];
    }
    return PHYLO_OK;
}

static int debug_device_lists_run(phylo_ctx* c, int32_t* lists, int64_t n_lists, int32_t* meta, int n_meta);

int phylo_debug_device_lists_of(phylo_ctx* c, const int64_t* ancestors, const int32_t* child, int32_t* lists, int64_t n_lists,
                                int32_t* meta, int n_meta) {
    CHK(bind(c));
    if (c->Kloc != c->K || c->K > PG_DL_MAX_K || !child || (c->N > 2 && !ancestors))
        return fail(c, PHYLO_EINVAL, "phylo_debug_device_lists_of: not sharded, K <= %d, ancestors and child given", PG_DL_MAX_K);
    CHK(ensure_sweep_state(c));
    CHK(ensure_graph_state(c));
    const int R = c->N - 1, K = c->K;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->swept = false;                                      // the sweep's genealogy is overwritten: no reverse pass on it after this
    c->last_graph = false;
    if (R > 1) HIPCHK(c, hipMemcpy(c->d_anc, ancestors, (size_t)(R - 1) * K * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_child, child, (size_t)R * K * 2 * 4, hipMemcpyHostToDevice));
    return debug_device_lists_run(c, lists, n_lists, meta, n_meta);
}

int phylo_debug_device_lists(phylo_ctx* c, int32_t* lists, int64_t n_lists, int32_t* meta, int n_meta, int64_t* ancestors, int32_t* child) {
    CHK(bind(c));
    if (!c->swept || !c->last_graph || !c->last_graph_marks || c->last_graph_twist || c->Kloc != c->K || c->K > PG_DL_MAX_K)
        return fail(c, PHYLO_ESTATE, "phylo_debug_device_lists needs a preceding lazy sweep with PHYLO_KEEP_GRAPH and the plain proposal, not sharded, K <= %d", PG_DL_MAX_K);
    const int R = c->N - 1, K = c->K;
    CHK(debug_device_lists_run(c, lists, n_lists, meta, n_meta));
    if (ancestors && R > 1) memcpy(ancestors, c->h_anc_p, (size_t)(R - 1) * K * 8);   // the pinned copies the sweep left
    if (child) memcpy(child, c->h_child_p, (size_t)R * K * 2 * 4);
    return PHYLO_OK;
}

static int debug_device_lists_run(phylo_ctx* c, int32_t* lists, int64_t n_lists, int32_t* meta, int n_meta) {
    const int R = c->N - 1, K = c->K;
    const size_t ints = pg_lists_ints((size_t)R, (size_t)K);
    if (!lists || !meta || n_lists < (int64_t)ints || n_meta < 6 + 3 * (R + 1))
        return fail(c, PHYLO_EINVAL, "phylo_debug_device_lists: lists needs %zu ints, meta %d", ints, 6 + 3 * (R + 1));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ad_off, 0xff, ints * 4, c->stream));        // (what the builders do not write stays -1)
    CHK(dev_lists_launch(c, c->stream, c->stream));
    dl_meta m;
    CHK(dev_lists_wait(c, m));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(lists, c->d_ad_off, ints * 4, hipMemcpyDeviceToHost));
    meta[0] = m.n_adp; meta[1] = m.n_chunks; meta[2] = 0; meta[3] = m.n_slow; meta[4] = m.n_par;
    meta[5] = (int32_t)pg_lists_cap((size_t)R, (size_t)K);
    for (int r = 0; r <= R; ++r) {
        meta[6 + r] = m.ev_adp0[r];
        meta[6 + (R + 1) + r] = 0;
        meta[6 + 2 * (R + 1) + r] = m.ev_slow0[r];
    }
    return PHYLO_OK;
}

int phylo_math_probe(phylo_ctx* c, int op, const double* x, const double* y, int n, double* out) {
    CHK(bind(c));
    if (n < 0 || (n > 0 && (!x || !y || !out))) return fail(c, PHYLO_EINVAL, "bad arguments");
    if (n == 0) return PHYLO_OK;
    void *dx, *dy, *dout;
    CHK(scratch_get(c, 0, (size_t)n * 8, &dx));
    CHK(scratch_get(c, 1, (size_t)n * 8, &dy));
    CHK(scratch_get(c, 2, (size_t)n * 8, &dout));
    HIPCHK(c, hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dy, y, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pk_math_probe, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, op, (const double*)dx, (const double*)dy, n,
                       (double*)dout);
    CHK(launch_check(c, "pk_math_probe"));
    HIPCHK(c, hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

// ---- multi-GPU --------------------------------------------------------------------------------
int phylo_comm_unique_id(char id[PHYLO_COMM_ID_BYTES]) {
    std::string err;
    int rc = phylo_comm_make_id(id, &err);
    if (rc != PHYLO_OK) g_last_error = err;
    return rc;
}

int phylo_comm_init(phylo_ctx* c, int rank, int world, const char id[PHYLO_COMM_ID_BYTES]) {
    CHK(bind(c));
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(c, PHYLO_EINVAL, "bad rank/world");
    if (c->K % world != 0) return fail(c, PHYLO_EINVAL, "K = %d is not divisible by world = %d", c->K, world);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_sweep_state(c);                                   // peers may still map the old pool: drop it first
    int rc = phylo_comm_setup(&c->comm, rank, world, id, &c->err);
    if (rc != PHYLO_OK) { g_last_error = c->err; return rc; }
    c->rank = rank;
    c->world = world;
    c->Kloc = c->K / world;
    c->k0 = rank * c->Kloc;
    c->swept = false;
    c->state_ready = false;
    CHK(alloc_sweep_state(c));                             // collective: every rank maps every peer's pool here
    CHK(refresh_leaf_ll(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_comm_share(phylo_ctx* c, phylo_ctx* owner) {
    CHK(bind(c));
    if (!owner || owner == c) return fail(c, PHYLO_EINVAL, "phylo_comm_share needs another context as the owner");
    if (owner->comm.parent) return fail(c, PHYLO_EINVAL, "the owner must hold its own communicator (phylo_comm_init)");
    if (owner->device != c->device) return fail(c, PHYLO_EINVAL, "both contexts must live on the same device");
    if (c->K % owner->world != 0) return fail(c, PHYLO_EINVAL, "K = %d is not divisible by world = %d", c->K, owner->world);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_sweep_state(c);
    phylo_comm_destroy(&c->comm);
    if (owner->comm.transport == 1 && !owner->comm.cstream)
        HIPCHK(c, hipStreamCreateWithFlags(&owner->comm.cstream, hipStreamNonBlocking));
    c->comm.parent = &owner->comm;
    c->comm.rank = owner->comm.rank;
    c->comm.world = owner->comm.world;
    c->comm.transport = owner->comm.transport;
    c->rank = owner->rank;
    c->world = owner->world;
    c->Kloc = c->K / c->world;
    c->k0 = c->rank * c->Kloc;
    c->swept = false;
    c->state_ready = false;
    CHK(alloc_sweep_state(c));                             // collective: every rank maps every peer's pool here
    CHK(refresh_leaf_ll(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PHYLO_OK;
}

int phylo_debug_remote_cache(phylo_ctx* c, int* used, int* cap) {
    CHK(bind(c));
    if (!used || !cap) return fail(c, PHYLO_EINVAL, "phylo_debug_remote_cache: NULL argument");
    *used = 0;
    *cap = c->cache_cap;
    if (c->d_mirror) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int32_t n = 0;
        HIPCHK(c, hipMemcpy(&n, c->d_mirror + (size_t)(c->N - 1) * c->K, 4, hipMemcpyDeviceToHost));
        *used = n;
    }
    return PHYLO_OK;
}

int phylo_comm_exchange_kind(const phylo_ctx* c) {
    if (!c || c->comm.transport == 0) return 0;
    return c->p2p ? 3 : c->comm.transport;
}

int phylo_comm_max(phylo_ctx* c, double* value) {
    CHK(bind(c));
    if (!value) return fail(c, PHYLO_EINVAL, "value is NULL");
    return phylo_comm_allreduce_max(c->comm, value, c->stream, &c->err);
}

int phylo_comm_allgather(phylo_ctx* c, const void* mine, size_t bytes, void* all) {
    CHK(bind(c));
    if (!mine || !all || bytes == 0) return fail(c, PHYLO_EINVAL, "bad arguments to phylo_comm_allgather");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return phylo_comm_allgather_host(c->comm, mine, bytes, all, c->stream, &c->err);
}

int phylo_comm_barrier(phylo_ctx* c) {
    double v = 0.0;
    return phylo_comm_max(c, &v);
}

}  // extern "C"
