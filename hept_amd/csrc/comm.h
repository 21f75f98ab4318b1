// RCCL communicator of the table-sharded operator: one process per GPU, one exchange step over xGMI.
// RCCL is bound at run time (dlopen of librccl.so.1 -- inside a PyTorch process that resolves to the copy torch has
// already loaded), so libhept_hip.so itself has no link-time dependency on it and loads in a plain C host.
#pragma once
#include "common.h"

struct hept_comm {
    void* nccl = nullptr;        // ncclComm_t
    int rank = 0, world = 1, device = 0;
    hipStream_t side = nullptr;  // transfers of finished head groups run here, behind the block attention
    hipEvent_t fork[HEPT_MAX_HEAD_GROUPS] = {};
    hipEvent_t join = nullptr;
};

// all-to-all of `bytes_per_peer` bytes per rank pair on `st` (ncclAllToAll on bytes)
int hept_comm_all_to_all(hept_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t st);
// in-place all-gather: rank r's `count` floats already sit at buf + r * count
int hept_comm_all_gather_f32(hept_comm* c, float* buf, size_t count, hipStream_t st);
void hept_comm_set_error(const char* what, const char* detail);
