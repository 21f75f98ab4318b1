// hept_comm: the RCCL side of table sharding (include/hept_hip.h, "Table sharding over the GPUs of one node").
// Nothing like it exists in the reference (single process, single device); the coupling it serves is the one line
// out = o.sum(0) / logits.sum(0), example/hept.py:79.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

std::mutex g_mu;
Rccl g_rccl;
thread_local std::string g_err;

template <class F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

// librccl.so.1: first the copy that is already mapped (PyTorch's), then the loader's search path, then /opt/rocm
Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_rccl.ok) return &g_rccl;
    if (g_rccl.handle) return nullptr;  // tried before, symbols missing
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        hept_comm_set_error("dlopen(librccl.so.1)", dlerror());
        return nullptr;
    }
    g_rccl.handle = h;
    const bool all = bind(h, "ncclGetUniqueId", g_rccl.GetUniqueId) && bind(h, "ncclCommInitRank", g_rccl.CommInitRank) &&
                     bind(h, "ncclCommDestroy", g_rccl.CommDestroy) && bind(h, "ncclAllToAll", g_rccl.AllToAll) &&
                     bind(h, "ncclAllGather", g_rccl.AllGather) && bind(h, "ncclGetErrorString", g_rccl.GetErrorString);
    if (!all) {
        hept_comm_set_error("dlsym", "librccl.so.1 lacks one of ncclGetUniqueId/CommInitRank/CommDestroy/AllToAll/AllGather");
        return nullptr;
    }
    g_rccl.ok = true;
    return &g_rccl;
}

int nccl_rc(Rccl* r, ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return HEPT_OK;
    hept_comm_set_error(what, r->GetErrorString ? r->GetErrorString(rc) : "?");
    return HEPT_ERR_COMM;
}

}  // namespace

void hept_comm_set_error(const char* what, const char* detail) {
    g_err = std::string(what ? what : "") + ": " + (detail ? detail : "");
}

extern "C" const char* hept_comm_last_error(void) { return g_err.c_str(); }

extern "C" int hept_comm_unique_id(void* id128) {
    if (!id128) return HEPT_ERR_ARG;
    Rccl* r = rccl();
    if (!r) return HEPT_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == HEPT_COMM_ID_BYTES, "unique id size");
    return nccl_rc(r, r->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

extern "C" int hept_comm_create(const void* id128, int rank, int world, hept_comm** out) {
    if (!id128 || !out || world < 1 || world > HEPT_MAX_RANKS || rank < 0 || rank >= world) return HEPT_ERR_ARG;
    Rccl* r = rccl();
    if (!r) return HEPT_ERR_COMM;
    hept_comm* c = new hept_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return HEPT_ERR_LAUNCH;
    }
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    int rc = nccl_rc(r, r->CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    if (rc) {
        delete c;
        return rc;
    }
    c->nccl = comm;
    if (hept_comm_init_streams(c)) {
        hept_comm_destroy(c);
        return HEPT_ERR_LAUNCH;
    }
    *out = c;
    return HEPT_OK;
}

int hept_comm_init_streams(hept_comm* c) {
    bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < HEPT_MAX_HEAD_GROUPS; ++i)
        ok = hipEventCreateWithFlags(&c->fork[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->join, hipEventDisableTiming) == hipSuccess;
    return ok ? HEPT_OK : HEPT_ERR_LAUNCH;
}

// a communicator without RCCL: rank bookkeeping, side stream and the one-sided transport only
extern "C" int hept_comm_create_local(int rank, int world, hept_comm** out) {
    if (!out || world < 1 || world > HEPT_MAX_RANKS || rank < 0 || rank >= world) return HEPT_ERR_ARG;
    hept_comm* c = new hept_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess || hept_comm_init_streams(c)) {
        hept_comm_destroy(c);
        return HEPT_ERR_LAUNCH;
    }
    *out = c;
    return HEPT_OK;
}

extern "C" int hept_comm_destroy(hept_comm* c) {
    if (!c) return HEPT_OK;
    // Everything this object owns belongs to c->device, and kernels of the caller's stream may still be storing into
    // the peers' mappings or using d_state (the Python owner can be collected right after an asynchronous forward):
    // make that device current and drain it before anything is unmapped or freed.
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    hept_p2p_release(c);
    for (int i = 0; i < HEPT_MAX_HEAD_GROUPS; ++i)
        if (c->fork[i]) (void)hipEventDestroy(c->fork[i]);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->side) (void)hipStreamDestroy(c->side);
    Rccl* r = rccl();
    if (r && c->nccl) (void)r->CommDestroy(static_cast<ncclComm_t>(c->nccl));
    delete c;
    if (prev >= 0) (void)hipSetDevice(prev);
    return HEPT_OK;
}

extern "C" int hept_comm_rank(const hept_comm* c) { return c ? c->rank : -1; }
extern "C" int hept_comm_world(const hept_comm* c) { return c ? c->world : -1; }

extern "C" int hept_comm_has_rccl(const hept_comm* c) { return c && c->nccl ? 1 : 0; }

int hept_comm_all_to_all(hept_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t st) {
    Rccl* r = rccl();
    if (!r || !c || !c->nccl) return HEPT_ERR_COMM;
    return nccl_rc(r, r->AllToAll(send, recv, bytes_per_peer, ncclUint8, static_cast<ncclComm_t>(c->nccl), st), "ncclAllToAll");
}

int hept_comm_all_gather_f32(hept_comm* c, float* buf, size_t count, hipStream_t st) {
    Rccl* r = rccl();
    if (!r || !c || !c->nccl) return HEPT_ERR_COMM;
    return nccl_rc(r, r->AllGather(buf + (size_t)c->rank * count, buf, count, ncclFloat, static_cast<ncclComm_t>(c->nccl), st),
                   "ncclAllGather");
}
