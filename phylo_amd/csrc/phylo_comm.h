// phylo_comm.h -- multi-rank plumbing of the sharded sweep: one process per GPU, particles sharded by
// contiguous ranges, global resampling.
//
// What crosses GPUs (DESIGN.md "Multi-GPU"):
//   * per rank event ONE grouped RCCL all-gather of three K-vectors (log-weights, log-likelihoods, node
//     log-likelihoods): 3 x 8 x K bytes, latency-bound, over xGMI;
//   * partial-likelihood vectors of nodes owned by another rank are READ IN PLACE over xGMI by the merge
//     kernel through peer mappings of every rank's node pool (hipIpc handles exchanged once at
//     phylo_comm_init); nothing is copied ahead of time, and only children that are actually merged move.
//
// Transports: RCCL (the product path) and "hostshm" (PHYLO_COMM=hostshm), a host-mediated all-gather
// through POSIX shared memory that exists so the sharded sweep -- bookkeeping, node addressing, peer pool
// mappings -- can be exercised by several processes on ONE GPU, where RCCL refuses duplicate devices.
// hostshm moves the same bytes the collective would; it computes nothing.
#pragma once
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/phylo_hip.h"

struct phylo_shm_header {
    std::atomic<unsigned int> arrived;
    std::atomic<unsigned int> generation;
    unsigned int world;
    unsigned int pad;
};

struct phylo_comm {
    int rank = 0, world = 1;
    int transport = 0;                 // 0 none (single rank), 1 RCCL, 2 hostshm
    ncclComm_t nccl = nullptr;
    // hostshm
    std::string shm_name;
    void* shm = nullptr;
    size_t shm_bytes = 0, slot_bytes = 0;
    // peer pools
    std::vector<void*> peer_base;      // opened IPC mappings (nullptr for self)
    std::vector<void*> peer_extra;     // further mapped buffers of the peers (root tables)
    // several contexts of one process on ONE communicator (phylo_comm_share): the sharers point at the owner; every
    // collective of the process then runs on the owner's dedicated stream, in host issue order, so that all ranks
    // see one communicator used from one stream in one order however many sweeps are in flight
    phylo_comm* parent = nullptr;
    hipStream_t cstream = nullptr;     // owner only, created by the first phylo_comm_share
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    double* bounce = nullptr;          // owner only: device bounce buffer of the small host exchanges, grown on demand
    size_t bounce_doubles = 0;
};

inline phylo_comm& phylo_comm_link(phylo_comm& c) { return c.parent ? *c.parent : c; }

#define PHYLO_SHM_SLOT (4u << 20)      // per-rank slot of the host-mediated transport

inline int phylo_comm_fail(std::string* err, const char* what, const char* detail) {
    if (err) *err = std::string(what) + ": " + (detail ? detail : "");
    return PHYLO_ECOMM;
}

inline int phylo_comm_transport_from_env() {
    const char* e = getenv("PHYLO_COMM");
    return (e && strcmp(e, "hostshm") == 0) ? 2 : 1;
}

inline int phylo_comm_make_id(char id[PHYLO_COMM_ID_BYTES], std::string* err) {
    memset(id, 0, PHYLO_COMM_ID_BYTES);
    if (phylo_comm_transport_from_env() == 2) {
        snprintf(id, PHYLO_COMM_ID_BYTES, "/phylo_shm_%d_%ld", (int)getpid(),
                 (long)std::chrono::steady_clock::now().time_since_epoch().count());
        return PHYLO_OK;
    }
    ncclUniqueId uid;
    ncclResult_t r = ncclGetUniqueId(&uid);
    if (r != ncclSuccess) return phylo_comm_fail(err, "ncclGetUniqueId", ncclGetErrorString(r));
    static_assert(sizeof(ncclUniqueId) <= PHYLO_COMM_ID_BYTES, "RCCL id does not fit");
    memcpy(id, &uid, sizeof(uid));
    return PHYLO_OK;
}

inline int phylo_shm_barrier(phylo_comm* c, std::string* err) {
    phylo_shm_header* h = (phylo_shm_header*)c->shm;
    const unsigned int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned int)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
        return PHYLO_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
            return phylo_comm_fail(err, "hostshm barrier", "timed out after 120 s (a rank died?)");
    }
    return PHYLO_OK;
}

inline void phylo_comm_destroy(phylo_comm* c) {
    for (void* p : c->peer_base)
        if (p) (void)hipIpcCloseMemHandle(p);
    c->peer_base.clear();
    for (void* p : c->peer_extra)
        if (p) (void)hipIpcCloseMemHandle(p);
    c->peer_extra.clear();
    if (c->ev_in) { (void)hipEventDestroy(c->ev_in); c->ev_in = nullptr; }
    if (c->ev_out) { (void)hipEventDestroy(c->ev_out); c->ev_out = nullptr; }
    if (c->parent) {                   // a sharer owns nothing else
        c->parent = nullptr;
        c->transport = 0; c->world = 1; c->rank = 0;
        return;
    }
    if (c->cstream) { (void)hipStreamDestroy(c->cstream); c->cstream = nullptr; }
    if (c->bounce) { (void)hipFree(c->bounce); c->bounce = nullptr; c->bounce_doubles = 0; }
    if (c->nccl) { (void)ncclCommDestroy(c->nccl); c->nccl = nullptr; }
    if (c->shm) {
        munmap(c->shm, c->shm_bytes);
        c->shm = nullptr;
        if (c->rank == 0) shm_unlink(c->shm_name.c_str());
    }
    c->transport = 0;
    c->world = 1;
    c->rank = 0;
}

inline int phylo_comm_setup(phylo_comm* c, int rank, int world, const char* id, std::string* err) {
    phylo_comm_destroy(c);
    c->rank = rank;
    c->world = world;
    if (world == 1 && !getenv("PHYLO_COMM_FORCE_RCCL")) { c->transport = 0; return PHYLO_OK; }
    c->transport = phylo_comm_transport_from_env();
    if (c->transport == 1) {
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof(uid));
        ncclResult_t r = ncclCommInitRank(&c->nccl, world, uid, rank);
        if (r != ncclSuccess) { c->nccl = nullptr; return phylo_comm_fail(err, "ncclCommInitRank", ncclGetErrorString(r)); }
        return PHYLO_OK;
    }
    // hostshm: rank 0 creates the segment, the others attach (the name arrives after creation: rank 0 only
    // publishes the id once phylo_comm_unique_id returned, and creates the segment here before its first barrier)
    c->shm_name.assign(id, strnlen(id, PHYLO_COMM_ID_BYTES));
    c->slot_bytes = PHYLO_SHM_SLOT;
    c->shm_bytes = 4096 + (size_t)world * c->slot_bytes;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        fd = shm_open(c->shm_name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) return phylo_comm_fail(err, "shm_open/ftruncate", c->shm_name.c_str());
    } else {
        while ((fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600)) < 0) {
            usleep(1000);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return phylo_comm_fail(err, "shm_open", "timed out");
        }
        // wait until rank 0 has sized the segment
        for (;;) {
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)c->shm_bytes) break;
            usleep(1000);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return phylo_comm_fail(err, "shm size", "timed out");
        }
    }
    c->shm = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->shm == MAP_FAILED) { c->shm = nullptr; return phylo_comm_fail(err, "mmap", c->shm_name.c_str()); }
    phylo_shm_header* h = (phylo_shm_header*)c->shm;
    if (rank == 0) h->world = (unsigned int)world;      // fresh segments are zero-filled: counters start at 0
    return phylo_shm_barrier(c, err);
}

// In-place all-gather of `n_arrays` device arrays of `count` doubles each (count = per-rank elements; array i
// holds world*count elements, this rank's at offset rank*count).
inline int phylo_comm_allgather_inplace(phylo_comm& me, double* const* arrays, int n_arrays, size_t count,
                                        hipStream_t stream, std::string* err) {
    if (me.transport == 0) return PHYLO_OK;
    phylo_comm& c = phylo_comm_link(me);
    if (c.transport == 1) {
        hipStream_t cs = stream;
        if (c.cstream) {               // shared communicator: hop onto its stream and back
            if (!me.ev_in && (hipEventCreateWithFlags(&me.ev_in, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&me.ev_out, hipEventDisableTiming) != hipSuccess))
                return phylo_comm_fail(err, "hipEventCreate", "shared communicator");
            if (hipEventRecord(me.ev_in, stream) != hipSuccess || hipStreamWaitEvent(c.cstream, me.ev_in, 0) != hipSuccess)
                return phylo_comm_fail(err, "hipStreamWaitEvent", "shared communicator (in)");
            cs = c.cstream;
        }
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n_arrays && r == ncclSuccess; ++i)
            r = ncclAllGather(arrays[i] + (size_t)c.rank * count, arrays[i], count, ncclDouble, c.nccl, cs);
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return phylo_comm_fail(err, "ncclAllGather", ncclGetErrorString(r != ncclSuccess ? r : r2));
        if (c.cstream && (hipEventRecord(me.ev_out, cs) != hipSuccess || hipStreamWaitEvent(stream, me.ev_out, 0) != hipSuccess))
            return phylo_comm_fail(err, "hipStreamWaitEvent", "shared communicator (out)");
        return PHYLO_OK;
    }
    const size_t bytes = count * sizeof(double), total = bytes * n_arrays;
    if (total > c.slot_bytes) return phylo_comm_fail(err, "hostshm all-gather", "message exceeds the slot size");
    char* slots = (char*)c.shm + 4096;
    if (hipStreamSynchronize(stream) != hipSuccess) return phylo_comm_fail(err, "hipStreamSynchronize", "hostshm");
    for (int i = 0; i < n_arrays; ++i)
        if (hipMemcpy(slots + (size_t)c.rank * c.slot_bytes + i * bytes, arrays[i] + (size_t)c.rank * count, bytes,
                      hipMemcpyDeviceToHost) != hipSuccess)
            return phylo_comm_fail(err, "hipMemcpy D2H", "hostshm");
    int rc = phylo_shm_barrier(&c, err);
    if (rc != PHYLO_OK) return rc;
    for (int p = 0; p < c.world; ++p) {
        if (p == c.rank) continue;
        for (int i = 0; i < n_arrays; ++i)
            if (hipMemcpy(arrays[i] + (size_t)p * count, slots + (size_t)p * c.slot_bytes + i * bytes, bytes,
                          hipMemcpyHostToDevice) != hipSuccess)
                return phylo_comm_fail(err, "hipMemcpy H2D", "hostshm");
    }
    return phylo_shm_barrier(&c, err);
}

// The same all-gather for several contexts that share one communicator, as ONE grouped collective: arrays holds
// n_arrays pointers per context, counts[i] = per-rank elements of context i's arrays.  RCCL: every context's stream
// is joined into the communicator's stream, one ncclGroup carries all the all-gathers, every stream waits for it.
inline int phylo_comm_allgather_group(phylo_comm* const* comms, double* const* arrays, int n_arrays, const size_t* counts,
                                      const hipStream_t* streams, int n_ctx, std::string* err) {
    phylo_comm& c = phylo_comm_link(*comms[0]);
    if (c.transport != 1 || !c.cstream) {                  // host-mediated transport or no shared stream: one by one
        for (int i = 0; i < n_ctx; ++i) {
            int rc = phylo_comm_allgather_inplace(*comms[i], arrays + (size_t)i * n_arrays, n_arrays, counts[i], streams[i], err);
            if (rc != PHYLO_OK) return rc;
        }
        return PHYLO_OK;
    }
    for (int i = 0; i < n_ctx; ++i) {
        phylo_comm& me = *comms[i];
        if (!me.ev_in && (hipEventCreateWithFlags(&me.ev_in, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&me.ev_out, hipEventDisableTiming) != hipSuccess))
            return phylo_comm_fail(err, "hipEventCreate", "shared communicator");
        if (hipEventRecord(me.ev_in, streams[i]) != hipSuccess || hipStreamWaitEvent(c.cstream, me.ev_in, 0) != hipSuccess)
            return phylo_comm_fail(err, "hipStreamWaitEvent", "grouped all-gather (in)");
    }
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < n_ctx && r == ncclSuccess; ++i)
        for (int a = 0; a < n_arrays && r == ncclSuccess; ++a) {
            double* p = arrays[(size_t)i * n_arrays + a];
            r = ncclAllGather(p + (size_t)c.rank * counts[i], p, counts[i], ncclDouble, c.nccl, c.cstream);
        }
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess)
        return phylo_comm_fail(err, "ncclAllGather (group)", ncclGetErrorString(r != ncclSuccess ? r : r2));
    phylo_comm& first = *comms[0];
    if (hipEventRecord(first.ev_out, c.cstream) != hipSuccess) return phylo_comm_fail(err, "hipEventRecord", "grouped all-gather (out)");
    for (int i = 0; i < n_ctx; ++i)
        if (hipStreamWaitEvent(streams[i], first.ev_out, 0) != hipSuccess)
            return phylo_comm_fail(err, "hipStreamWaitEvent", "grouped all-gather (out)");
    return PHYLO_OK;
}

// device bounce buffer of the communicator (allocated once, grown when a larger exchange comes)
inline int phylo_comm_bounce(phylo_comm& link, size_t doubles, double** out, std::string* err) {
    if (link.bounce_doubles < doubles) {
        if (link.bounce) (void)hipFree(link.bounce);
        link.bounce = nullptr;
        link.bounce_doubles = 0;
        const size_t want = doubles < 1024 ? 1024 : doubles;
        if (hipMalloc((void**)&link.bounce, want * 8) != hipSuccess) return phylo_comm_fail(err, "hipMalloc", "bounce buffer");
        link.bounce_doubles = want;
    }
    *out = link.bounce;
    return PHYLO_OK;
}

// all-gather of small host blobs (IPC handles, per-particle outputs) through the communicator's device bounce buffer
inline int phylo_comm_allgather_host(phylo_comm& c, const void* mine, size_t bytes, void* all, hipStream_t stream,
                                     std::string* err) {   // c may be a sharer: rank/world are mirrored, the link is resolved below
    if (c.transport == 0) { memcpy(all, mine, bytes); return PHYLO_OK; }
    const size_t count = (bytes + 7) / 8;
    double* d = nullptr;
    int rc = phylo_comm_bounce(phylo_comm_link(c), count * c.world, &d, err);
    if (rc != PHYLO_OK) return rc;
    std::vector<double> tmp(count, 0.0);
    memcpy(tmp.data(), mine, bytes);
    if (hipMemcpy(d + (size_t)c.rank * count, tmp.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = phylo_comm_fail(err, "hipMemcpy", "bounce H2D");
    double* arr[1] = {d};
    if (rc == PHYLO_OK) rc = phylo_comm_allgather_inplace(c, arr, 1, count, stream, err);
    if (rc == PHYLO_OK && hipStreamSynchronize(stream) != hipSuccess) rc = phylo_comm_fail(err, "hipStreamSynchronize", "bounce");
    std::vector<double> out(count * c.world);
    if (rc == PHYLO_OK && hipMemcpy(out.data(), d, count * 8 * c.world, hipMemcpyDeviceToHost) != hipSuccess)
        rc = phylo_comm_fail(err, "hipMemcpy", "bounce D2H");
    if (rc == PHYLO_OK)
        for (int p = 0; p < c.world; ++p) memcpy((char*)all + (size_t)p * bytes, out.data() + (size_t)p * count, bytes);
    return rc;
}

// Exchange IPC handles of every rank's node pool and map the peers' pools.  ptrs_out[p] = device address of
// rank p's pool in THIS process (own pool for p == rank).
inline int phylo_comm_map_pools(phylo_comm& c, void* my_pool, std::vector<void*>* ptrs_out, hipStream_t stream,
                                std::string* err) {
    for (void* p : c.peer_base)
        if (p) (void)hipIpcCloseMemHandle(p);
    for (void* p : c.peer_extra)
        if (p) (void)hipIpcCloseMemHandle(p);
    c.peer_extra.clear();
    c.peer_base.assign(c.world, nullptr);
    ptrs_out->assign(c.world, nullptr);
    (*ptrs_out)[c.rank] = my_pool;
    if (c.world == 1) return PHYLO_OK;
    hipIpcMemHandle_t mine;
    hipError_t e = hipIpcGetMemHandle(&mine, my_pool);
    if (e != hipSuccess) return phylo_comm_fail(err, "hipIpcGetMemHandle", hipGetErrorString(e));
    std::vector<hipIpcMemHandle_t> all(c.world);
    int rc = phylo_comm_allgather_host(c, &mine, sizeof(mine), all.data(), stream, err);
    if (rc != PHYLO_OK) return rc;
    for (int p = 0; p < c.world; ++p) {
        if (p == c.rank) continue;
        void* base = nullptr;
        e = hipIpcOpenMemHandle(&base, all[p], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return phylo_comm_fail(err, "hipIpcOpenMemHandle", hipGetErrorString(e));
        c.peer_base[p] = base;
        (*ptrs_out)[p] = base;
    }
    return PHYLO_OK;
}

// Same exchange for one more buffer per rank (the root-table slab); call after phylo_comm_map_pools.
inline int phylo_comm_map_extra(phylo_comm& c, void* mine_dev, std::vector<void*>* ptrs_out, hipStream_t stream,
                                std::string* err) {
    ptrs_out->assign(c.world, nullptr);
    (*ptrs_out)[c.rank] = mine_dev;
    if (c.world == 1) return PHYLO_OK;
    hipIpcMemHandle_t mine;
    hipError_t e = hipIpcGetMemHandle(&mine, mine_dev);
    if (e != hipSuccess) return phylo_comm_fail(err, "hipIpcGetMemHandle", hipGetErrorString(e));
    std::vector<hipIpcMemHandle_t> all(c.world);
    int rc = phylo_comm_allgather_host(c, &mine, sizeof(mine), all.data(), stream, err);
    if (rc != PHYLO_OK) return rc;
    for (int p = 0; p < c.world; ++p) {
        if (p == c.rank) continue;
        void* base = nullptr;
        e = hipIpcOpenMemHandle(&base, all[p], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return phylo_comm_fail(err, "hipIpcOpenMemHandle", hipGetErrorString(e));
        c.peer_extra.push_back(base);
        (*ptrs_out)[p] = base;
    }
    return PHYLO_OK;
}

// max over ranks of a host double (also a barrier).  RCCL: an all-reduce (ncclMax) of one device double on the
// communicator's stream; host-mediated test transport: all-gather + host max.
inline int phylo_comm_allreduce_max(phylo_comm& me, double* value, hipStream_t stream, std::string* err) {
    if (me.transport == 0) return PHYLO_OK;
    phylo_comm& c = phylo_comm_link(me);
    if (c.transport == 1) {
        double* d = nullptr;
        int rc = phylo_comm_bounce(c, 2, &d, err);
        if (rc != PHYLO_OK) return rc;
        hipStream_t cs = c.cstream ? c.cstream : stream;
        if (hipStreamSynchronize(stream) != hipSuccess) return phylo_comm_fail(err, "hipStreamSynchronize", "all-reduce");
        if (hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, cs) != hipSuccess) return phylo_comm_fail(err, "hipMemcpyAsync", "all-reduce in");
        ncclResult_t r = ncclAllReduce(d, d + 1, 1, ncclDouble, ncclMax, c.nccl, cs);
        if (r != ncclSuccess) return phylo_comm_fail(err, "ncclAllReduce", ncclGetErrorString(r));
        if (hipMemcpyAsync(value, d + 1, 8, hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess)
            return phylo_comm_fail(err, "hipMemcpyAsync", "all-reduce out");
        return PHYLO_OK;
    }
    std::vector<double> all(c.world);
    int rc = phylo_comm_allgather_host(me, value, sizeof(double), all.data(), stream, err);
    if (rc != PHYLO_OK) return rc;
    double m = all[0];
    for (int p = 1; p < c.world; ++p) m = all[p] > m ? all[p] : m;
    *value = m;
    return PHYLO_OK;
}
