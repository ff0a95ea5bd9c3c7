// phylo_comm.h -- multi-rank plumbing of the sharded sweep (one process per GPU).  Placeholder for the
// single-GPU milestone: world == 1 is the identity; world > 1 is refused until the RCCL path lands.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/phylo_hip.h"

struct phylo_comm {
    int rank = 0, world = 1;
};

inline void phylo_comm_destroy(phylo_comm*) {}

inline int phylo_comm_make_id(char id[PHYLO_COMM_ID_BYTES], std::string* err) {
    (void)err;
    for (int i = 0; i < PHYLO_COMM_ID_BYTES; ++i) id[i] = 0;
    return PHYLO_OK;
}

inline int phylo_comm_setup(phylo_comm* c, int rank, int world, const char*, std::string* err) {
    if (world != 1) {
        *err = "multi-rank sweeps are not available in this build";
        return PHYLO_ECOMM;
    }
    c->rank = rank;
    c->world = world;
    return PHYLO_OK;
}

inline int phylo_comm_gather_rank(phylo_comm&, double*, double*, double*, int, hipStream_t, std::string*) {
    return PHYLO_OK;
}

inline int phylo_comm_allreduce_max(phylo_comm&, double*, hipStream_t, std::string*) { return PHYLO_OK; }
