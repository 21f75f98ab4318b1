// One-sided exchange of the table-sharded operator (the coupling it serves is the one line
// `out = o.sum(0) / logits.sum(0)`, example/hept.py:79).
//
// Every rank owns an exchange buffer in *uncached* device memory (peers write it while it is being polled, so it must
// never sit stale in an L2), exported as a HIP IPC handle and mapped by every other rank of the node:
//     [ flags 4 KiB | recv (groups, world, per, hg, row) | out (world * per, D) f32 | out' (the same again) ]
// Rows: the kernel that sums a rank's local tables stores each point's row straight into the buffer of the rank that
// finishes that point (xGMI stores, 16 B per lane, runs of 64-B rows); the NEXT kernel of the stream -- the one that
// is about to wait for the other ranks' rows -- raises flag[group][source] = epoch in every destination.  Output: the
// rank's finished (per, D) slice is stored into every rank's `out` region, and the gather kernel that follows raises
// flag_out[source] = epoch before it waits for the others'.  A flag therefore goes up only after every wave that stored
// the data it announces has ended (stores acknowledged: p2p_dev.h, raise_flags); the consumer polls its own (local)
// flags with system-scope acquire loads.  One step is in
// flight at a time: a rank leaves a step only when every rank's output slice has arrived, i.e. after every rank has
// finished reading the rows it received, so single buffers are enough.  (The second output region serves the view mode
// only -- hept_comm_set_out_view: the caller reads the gathered output in place, so step e + 1 must not write where
// step e's output lies; step e + 2 may, see include/hept_hip.h.)  Polling is bounded (20 s, once: the
// error is sticky): on a timeout the kernel records it in the communicator's status word and every later wait returns
// at once, so a lost peer cannot hang the GPU; the host reads the word with hept_comm_status.
#include "comm.h"
#include "p2p_dev.h"

#include <stdlib.h>

namespace {

template <bool P16>
__global__ __launch_bounds__(256) void reduce_push_kernel(PushArgs a) {
    reduce_push_body<P16>(a, blockIdx.x);
}

// (raises this rank's row flags first: the launches that stored its rows have ended -- p2p_dev.h raise_flags)
__global__ __launch_bounds__(256) void wait_rows_kernel(char* local, char* const* peers, int me, int head_groups, int world,
                                                        unsigned int epoch, unsigned int* status,
                                                        unsigned long long timeout) {
    const int i = threadIdx.x;
    if (peers) raise_flags(peers, world, head_groups, me, HEPT_MAX_RANKS, epoch);   // (null: the producers signalled)
    if (i < head_groups * world) {
        const int g = i / world, s = i - g * world;
        wait_flag(flag_word(local, g * HEPT_MAX_RANKS + s), epoch, status, 1u, timeout);
    }
}

// this rank's slice of the output (already in its own `out` region) -> the same place in every other rank's buffer
// (a slice computed after a timed-out wait goes out as NaN, see combine_out_kernel<PUSH>)
// (the output flag is raised by the gather kernel that follows)
__global__ __launch_bounds__(256) void push_out_kernel(char* const* peers, int world, int me, size_t slice_off,
                                                       size_t slice_bytes, const unsigned int* status,
                                                       unsigned int* counter, unsigned int epoch) {
    const u32x4* src = reinterpret_cast<const u32x4*>(peers[me] + slice_off);
    const size_t n16 = slice_bytes / 16;
    const bool poisoned = status_bad(status);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = src[i];
        if (poisoned) v = u32x4{0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u};
        for (int s = 0; s < world; ++s)
            if (s != me) store16_system(peers[s] + slice_off + i * 16, v);
    }
    drain_remote_stores();
    if (counter) signal_when_all_done(counter, peers, world, OUT_FLAG_WORD + me, epoch, gridDim.x);
}

// `bytes` = the whole (n_pad, D) output, `valid_bytes` = its first N rows: the padding rows are written as zeros (the
// RCCL and torch transports zero them too; nobody stores them in the fused combine).  If any wait of this step -- here
// or in an earlier kernel -- has timed out, the rows it announces may be unfinished: the WHOLE output is then written
// as NaN, so that a lost or slow peer can never turn into plausible numbers (the host also finds the status word at
// its next call, hept_forward_sharded returns HEPT_ERR_COMM from then on).
// [skip_lo, skip_hi): 16-B pieces the combine has already stored into dst (this rank's own slice); they are still
// overwritten on the failure path.
__global__ __launch_bounds__(256) void wait_copy_out_kernel(char* local, int world, unsigned int epoch, size_t out_off,
                                                            size_t bytes, size_t valid_bytes, float* __restrict__ dst,
                                                            unsigned int* status, unsigned long long timeout,
                                                            size_t skip_lo, size_t skip_hi, char* const* peers, int me) {
    // this rank's slice has been stored into every rank's output region by the kernel in front of this one
    // (peers null: that kernel raised the flag itself, HEPT_P2P_PRODUCER_SIGNAL)
    if (peers && blockIdx.x < RAISE_WGS) raise_flags(peers, world, 1, OUT_FLAG_WORD + me, 0, epoch);
    if (threadIdx.x < world) wait_flag(flag_word(local, OUT_FLAG_WORD + threadIdx.x), epoch, status, 2u, timeout);
    __syncthreads();
    const bool bad = status_bad(status);
    const u32x4* src = reinterpret_cast<const u32x4*>(local + out_off);
    u32x4* out = reinterpret_cast<u32x4*>(dst);
    const size_t n16 = bytes / 16, v16 = valid_bytes / 16, stride = (size_t)gridDim.x * blockDim.x;
    const u32x4 nan4 = {0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u}, zero4 = {0u, 0u, 0u, 0u};
    // uncached reads have a long latency: four pieces in flight per thread
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + u * stride;
            if (j < v16 && !bad && !(j >= skip_lo && j < skip_hi)) v[u] = src[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + u * stride;
            if (j < n16 && (bad || j >= v16 || !(j >= skip_lo && j < skip_hi))) out[j] = bad ? nan4 : (j < v16 ? v[u] : zero4);
        }
    }
}

// view mode: the waits of wait_copy_out_kernel without the copy.  On the good path nothing is written; after a timed-out
// wait the N * D floats the caller is about to read are overwritten with NaN, like the copied output would be.
__global__ __launch_bounds__(256) void wait_out_kernel(char* local, int world, unsigned int epoch, size_t out_off,
                                                       size_t valid_bytes, unsigned int* status,
                                                       unsigned long long timeout, char* const* peers, int me) {
    if (peers) raise_flags(peers, world, 1, OUT_FLAG_WORD + me, 0, epoch);
    if (threadIdx.x < world) wait_flag(flag_word(local, OUT_FLAG_WORD + threadIdx.x), epoch, status, 2u, timeout);
    __syncthreads();
    if (!status_bad(status)) return;
    const u32x4 nan4 = {0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < valid_bytes / 16; i += (size_t)gridDim.x * blockDim.x)
        store16_system(local + out_off + i * 16, nan4);
}

#ifndef HEPT_COPY_OUT_WGS
#define HEPT_COPY_OUT_WGS 256
#endif
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
// HEPT_P2P_PRODUCER_SIGNAL=1: the protocol of rounds 2-4 (p2p_dev.h signal_when_all_done) instead of "a kernel boundary is
// the completion signal" -- an A/B switch and a fallback for a stack on which a wave's stores to a peer GPU are not
// acknowledged before the wave retires.  Must be set on EVERY rank (it decides who raises the flags).
inline bool producer_signal() {
    static const bool on = [] { const char* e = getenv("HEPT_P2P_PRODUCER_SIGNAL"); return e && *e && *e != '0'; }();
    return on;
}
// completion counters in hept_comm::d_state: [0] row producers of a launch, [1] output producers
constexpr int STATE_ROW_COUNTER = 0, STATE_OUT_COUNTER = 1;
constexpr int CMB_WAIT_MAX = 256;  // threads of the combine that poll one arrival flag each

}  // namespace

P2pLayout hept_p2p_layout(int N, int H, int D, int world, int precision) {
    const size_t per = ((size_t)N + world - 1) / world, n_pad = per * world;
    const size_t row = hept_part_precision(precision, D) == HEPT_PREC_BF16 ? 64 : 128;
    P2pLayout l;
    l.recv_off = HEPT_P2P_FLAG_BYTES;
    l.out_off = l.recv_off + up256(n_pad * H * row);
    l.out_bytes = up256(n_pad * D * 4);
    l.bytes = l.out_off + 2 * l.out_bytes;
    return l;
}

extern "C" size_t hept_p2p_bytes(int N, int H, int D, int world, int precision) {
    if (N < 1 || H < 1 || D < 1 || world < 1) return 0;
    return hept_p2p_layout(N, H, D, world, precision).bytes;
}

void hept_p2p_release(hept_comm* c) {
    if (!c) return;
    for (int s = 0; s < HEPT_MAX_RANKS; ++s) {
        if (c->p2p_peer[s] && s != c->rank) (void)hipIpcCloseMemHandle(c->p2p_peer[s]);
        c->p2p_peer[s] = nullptr;
    }
    if (c->p2p_local) (void)hipFree(c->p2p_local);
    if (c->p2p_self) (void)hipFree(c->p2p_self);
    c->p2p_self = nullptr;
    if (c->d_peer) (void)hipFree(c->d_peer);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->h_status) (void)hipHostFree(c->h_status);
    c->p2p_local = nullptr;
    c->d_peer = nullptr;
    c->d_state = nullptr;
    c->h_status = nullptr;
    c->p2p_bytes = 0;
    c->p2p_open = false;
    c->last_out = nullptr;
}

// Allocate this rank's exchange buffer (uncached device memory) and export it.  Collective by convention: every rank
// allocates the same size, the HEPT_IPC_HANDLE_BYTES handles reach every rank by any host-side means, every rank
// calls hept_comm_p2p_open with all of them (rank order).
extern "C" int hept_comm_p2p_alloc(hept_comm* c, size_t bytes, void* handle_out) {
    if (!c || !handle_out || bytes < HEPT_P2P_FLAG_BYTES) return HEPT_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == HEPT_IPC_HANDLE_BYTES, "IPC handle size");
    if (hipDeviceSynchronize() != hipSuccess) return HEPT_ERR_LAUNCH;  // nothing in flight may still use the old buffers
    hept_p2p_release(c);
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            hept_comm_set_error("hipExtMallocWithFlags", "neither uncached nor fine-grained device memory is available");
            return HEPT_ERR_COMM;
        }
    }
    c->p2p_local = static_cast<char*>(p);
    c->p2p_bytes = bytes;
    bool ok = hipMemset(p, 0, HEPT_P2P_FLAG_BYTES) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&c->p2p_self), bytes) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&c->d_peer), sizeof(char*) * HEPT_MAX_RANKS) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&c->d_state), sizeof(unsigned int) * 32) == hipSuccess;
    ok = ok && hipMemset(c->d_state, 0, sizeof(unsigned int) * 32) == hipSuccess;
    // host-mapped copy of the status word (written by the device on a timeout only; read by the host before every step)
    void* hs = nullptr;
    ok = ok && hipHostMalloc(&hs, 64, hipHostMallocMapped) == hipSuccess;
    if (ok) {
        c->h_status = static_cast<unsigned int*>(hs);
        *c->h_status = 0u;
        void* dptr = nullptr;
        ok = hipHostGetDevicePointer(&dptr, hs, 0) == hipSuccess;
        const unsigned long long addr = reinterpret_cast<unsigned long long>(dptr);
        ok = ok && hipMemcpy(c->d_state + HEPT_STATE_HOSTPTR, &addr, sizeof(addr), hipMemcpyHostToDevice) == hipSuccess;
    }
    c->broken = false;
    hipIpcMemHandle_t h;
    __builtin_memset(&h, 0, sizeof(h));
    if (ok && c->world > 1 && hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError();
        hept_comm_set_error("hipIpcGetMemHandle", "cannot export the exchange buffer (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
        ok = false;
    }
    if (!ok) {
        hept_p2p_release(c);
        return HEPT_ERR_COMM;
    }
    __builtin_memcpy(handle_out, &h, sizeof(h));
    c->epoch = 0;
    // wall_clock64 ticks at the device's constant wall clock rate (kHz); HEPT_P2P_TIMEOUT_S seconds, default 20
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    double secs = 20.0;
    if (const char* e = getenv("HEPT_P2P_TIMEOUT_S")) secs = atof(e) > 0 ? atof(e) : secs;
    c->timeout_ticks = (unsigned long long)(secs * 1e3 * khz);
    return HEPT_OK;
}

extern "C" int hept_comm_p2p_open(hept_comm* c, const void* handles) {
    if (!c || !handles || !c->p2p_local) return HEPT_ERR_ARG;
    const char* hs = static_cast<const char*>(handles);
    for (int s = 0; s < c->world; ++s) {
        if (s == c->rank) {
            c->p2p_peer[s] = c->p2p_local;
            continue;
        }
        hipIpcMemHandle_t h;
        __builtin_memcpy(&h, hs + (size_t)s * sizeof(h), sizeof(h));
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            hept_comm_set_error("hipIpcOpenMemHandle", "cannot map a peer's exchange buffer");
            return HEPT_ERR_COMM;
        }
        c->p2p_peer[s] = static_cast<char*>(p);
    }
    if (hipMemcpy(c->d_peer, c->p2p_peer, sizeof(char*) * HEPT_MAX_RANKS, hipMemcpyHostToDevice) != hipSuccess)
        return HEPT_ERR_LAUNCH;
    // first launch out of this library in a fresh process loads its code object (seconds on a cold box): do it now,
    // not inside the first exchange, where the other ranks would be polling for this one
    hipLaunchKernelGGL(wait_rows_kernel, dim3(1), dim3(64), 0, nullptr, c->p2p_local, (char* const*)nullptr, c->rank, 0, c->world, 0u, c->d_state + HEPT_STATE_STATUS, 0ull);
    if (hipDeviceSynchronize() != hipSuccess) return HEPT_ERR_LAUNCH;
    c->p2p_open = true;
    return HEPT_OK;
}

extern "C" int hept_comm_p2p_ready(const hept_comm* c, size_t bytes) { return c && c->p2p_open && c->p2p_bytes >= bytes ? 1 : 0; }

// debugging aid: this rank's flag words (HEPT_P2P_FLAG_BYTES bytes) and its epoch
extern "C" int hept_comm_p2p_flags(hept_comm* c, void* out_flags, unsigned int* epoch) {
    if (!c || !out_flags || !epoch || !c->p2p_local) return HEPT_ERR_ARG;
    if (hipMemcpy(out_flags, c->p2p_local, HEPT_P2P_FLAG_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return HEPT_ERR_LAUNCH;
    *epoch = c->epoch;
    return HEPT_OK;
}

// 0 = healthy; bit 0: a wait for rows timed out, bit 1: a wait for the output timed out (synchronises the device)
extern "C" int hept_comm_status(hept_comm* c, int* status) {
    if (!c || !status) return HEPT_ERR_ARG;
    *status = 0;
    if (!c->d_state) return HEPT_OK;
    unsigned int v = 0;
    if (hipMemcpy(&v, c->d_state + HEPT_STATE_STATUS, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return HEPT_ERR_LAUNCH;
    *status = (int)v | (c->broken ? 4 : 0);   // bit 2: a step failed on the host after its epoch was taken
    return HEPT_OK;
}

// What the host can see without touching the device: non-zero once a device-side wait has timed out (the kernel that
// timed out wrote the host-mapped word) or a step failed after taking its epoch.  hept_forward_sharded checks it first.
int hept_p2p_failed(const hept_comm* c) {
    if (!c) return 0;
    unsigned int v = c->h_status ? __atomic_load_n(c->h_status, __ATOMIC_RELAXED) : 0u;
    return (int)v | (c->broken ? 4 : 0);
}

// Forget a recorded failure and restart the protocol: status words cleared, epoch and every arrival flag and completion
// counter back to zero.  COLLECTIVE by convention: after a failure the ranks' epochs may differ (a rank that failed
// before taking its epoch is one behind for good), so every rank calls this, and the ranks meet at a host barrier
// before the next exchange (TableSharding.check / tune do).  Synchronises the device.
extern "C" int hept_comm_reset_status(hept_comm* c) {
    if (!c) return HEPT_ERR_ARG;
    if (!c->d_state) return HEPT_OK;
    if (hipDeviceSynchronize() != hipSuccess) return HEPT_ERR_LAUNCH;
    bool ok = hipMemset(c->d_state, 0, sizeof(unsigned int) * HEPT_STATE_HOSTPTR) == hipSuccess;
    if (c->p2p_local) ok = ok && hipMemset(c->p2p_local, 0, HEPT_P2P_FLAG_BYTES) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (c->h_status) __atomic_store_n(c->h_status, 0u, __ATOMIC_RELAXED);
    c->broken = false;
    c->epoch = 0;
    return ok ? HEPT_OK : HEPT_ERR_LAUNCH;
}

int hept_p2p_push_args(hept_comm* c, const float* part, int part_precision, int Tl, int N, int H, int D, int h0, int hg,
                       int g, int acc_precision, const P2pLayout& lay, int push_wgs, bool mirror, PushArgs* out) {
    if (!c || !c->p2p_open || !part || !out) return HEPT_ERR_ARG;
    if (acc_precision != part_precision) return HEPT_ERR_SHAPE;  // the exchange keeps the row format of block_attn
    if (part_precision == HEPT_PREC_BF16 && D != 24) return HEPT_ERR_SHAPE;
    const int per = (N + c->world - 1) / c->world;
    const size_t row = acc_precision == HEPT_PREC_BF16 ? 64 : 128;
    PushArgs a;
    a.part = part;
    a.Tl = Tl; a.N = N; a.H = H; a.h0 = h0; a.hg = hg;
    a.per = per; a.world = c->world; a.me = c->rank;
    a.peers = c->d_peer;
    a.recv_off = lay.recv_off;
    a.group_off = (size_t)g * per * c->world * hg * row;
    a.epoch = c->epoch;
    a.flag_idx = g * HEPT_MAX_RANKS + c->rank;
    a.push_wgs = push_wgs;
    a.direct = 0;
    a.counter = producer_signal() ? c->d_state + STATE_ROW_COUNTER : nullptr;
    a.self = mirror ? c->p2p_self : c->p2p_local;
    *out = a;
    return HEPT_OK;
}

// one local table: the block-attention launch of head group g scatters its rows into the owners' buffers itself
int hept_p2p_direct_args(hept_comm* c, int N, int H, int D, int h0, int hg, int g, int acc_precision,
                         const P2pLayout& lay, bool mirror, PushArgs* out) {
    static const float dummy = 0.f;   // (hept_p2p_push_args wants a row pointer; direct launches never read it)
    int rc = hept_p2p_push_args(c, &dummy, acc_precision, 1, N, H, D, h0, hg, g, acc_precision, lay, 0, mirror, out);
    if (rc) return rc;
    out->part = nullptr;
    out->direct = 1;
    return HEPT_OK;
}

int hept_p2p_reduce_push(hept_comm* c, const float* part, int part_precision, int Tl, int N, int H, int D, int h0,
                         int hg, int g, int acc_precision, const P2pLayout& lay, bool mirror, hipStream_t st) {
    const size_t row = acc_precision == HEPT_PREC_BF16 ? 64 : 128;
    const size_t blocks = ((size_t)N * hg * (row / 16) + 255) / 256;
    PushArgs a;
    int rc = hept_p2p_push_args(c, part, part_precision, Tl, N, H, D, h0, hg, g, acc_precision, lay,
                                (int)(blocks < 4096 ? blocks : 4096), mirror, &a);
    if (rc) return rc;
    if (part_precision == HEPT_PREC_BF16)
        hipLaunchKernelGGL((reduce_push_kernel<true>), dim3(a.push_wgs), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((reduce_push_kernel<false>), dim3(a.push_wgs), dim3(256), 0, st, a);
    return hept_launch_status();
}

int hept_p2p_wait_rows(hept_comm* c, int head_groups, hipStream_t st) {
    if (!c || !c->p2p_open || head_groups * c->world > 256) return HEPT_ERR_ARG;
    hipLaunchKernelGGL(wait_rows_kernel, dim3(1), dim3(256), 0, st, c->p2p_local,
                       producer_signal() ? (char* const*)nullptr : (char* const*)c->d_peer, c->rank, head_groups, c->world,
                       c->epoch, c->d_state + HEPT_STATE_STATUS, c->timeout_ticks);
    return hept_launch_status();
}

int hept_p2p_push_out(hept_comm* c, int per, int D, const P2pLayout& lay, hipStream_t st) {
    if (!c || !c->p2p_open) return HEPT_ERR_ARG;
    const size_t slice_bytes = (size_t)per * D * 4;
    if (slice_bytes % 16 != 0) return HEPT_ERR_SHAPE;
    const size_t blocks = (slice_bytes / 16 + 255) / 256;
    hipLaunchKernelGGL(push_out_kernel, dim3((unsigned)(blocks < 512 ? (blocks ? blocks : 1) : 512)), dim3(256), 0, st,
                       c->d_peer, c->world, c->rank, lay.out_off + (size_t)c->rank * slice_bytes, slice_bytes,
                       c->d_state + HEPT_STATE_STATUS, producer_signal() ? c->d_state + STATE_OUT_COUNTER : nullptr, c->epoch);
    return hept_launch_status();
}

int hept_combine_push(const float* part, int part_precision, int Tl, int N, int H, int n_count, int HG,
                      size_t group_stride, const float* out_weight, const float* out_bias, const P2pDev& px,
                      hipStream_t st);   // combine.hip

int hept_p2p_combine_push(hept_comm* c, int head_groups, int per, int cnt, int H, int hg, int acc_precision,
                          const float* out_weight, const float* out_bias, const P2pLayout& lay, float* out_local,
                          hipStream_t st) {
    if (!c || !c->p2p_open || cnt < 1 || head_groups * c->world > CMB_WAIT_MAX) return HEPT_ERR_ARG;
    static_assert(HEPT_MAX_RANKS == HEPT_MAX_RANKS_DEV, "flag table pitch");
    const size_t row = acc_precision == HEPT_PREC_BF16 ? 64 : 128;
    P2pDev px;
    px.peers = c->d_peer;
    px.local = c->p2p_local;
    px.world = c->world;
    px.me = c->rank;
    px.epoch = c->epoch;
    px.status = c->d_state + HEPT_STATE_STATUS;
    px.timeout = c->timeout_ticks;
    px.wait_groups = head_groups;
    px.consumer_raises = producer_signal() ? 0 : 1;
    px.out_counter = producer_signal() ? c->d_state + STATE_OUT_COUNTER : nullptr;
    px.slice_off = lay.out_off + (size_t)c->rank * per * 24 * 4;
    px.self_rows = reinterpret_cast<const float*>(c->p2p_self + lay.recv_off);
    px.out_local = reinterpret_cast<char*>(out_local);
    px.out_base = lay.out_off;
    return hept_combine_push(reinterpret_cast<const float*>(c->p2p_local + lay.recv_off), acc_precision, c->world, per, H,
                             cnt, hg, (size_t)per * c->world * hg * row / 4, out_weight, out_bias, px, st);
}

int hept_p2p_wait_out(hept_comm* c, int N, int D, const P2pLayout& lay, hipStream_t st) {
    if (!c || !c->p2p_open) return HEPT_ERR_ARG;
    const size_t valid = (size_t)N * D * 4;
    if (valid % 16 != 0) return HEPT_ERR_SHAPE;
    // (every workgroup polls the same few flags: the extra ones are there for the NaN fill of the failure path)
    hipLaunchKernelGGL(wait_out_kernel, dim3(16), dim3(256), 0, st, c->p2p_local, c->world, c->epoch, lay.out_off, valid,
                       c->d_state + HEPT_STATE_STATUS, c->timeout_ticks,
                       producer_signal() ? (char* const*)nullptr : (char* const*)c->d_peer, c->rank);
    return hept_launch_status();
}

// Output gather without the copy.  COLLECTIVE by convention (every rank of the communicator sets the same mode before
// the next step: the mode decides which of the two output regions a step's slices are stored into).
extern "C" int hept_comm_set_out_view(hept_comm* c, int on) {
    if (!c) return HEPT_ERR_ARG;
    c->out_view = on != 0;
    c->last_out = nullptr;
    return HEPT_OK;
}

extern "C" int hept_comm_out_view(hept_comm* c, const float** out) {
    if (!c || !out) return HEPT_ERR_ARG;
    *out = c->last_out;
    return c->last_out ? HEPT_OK : HEPT_ERR_ARG;
}

int hept_p2p_wait_copy_out(hept_comm* c, int n_pad, int N, int D, const P2pLayout& lay, float* dst, bool skip_own,
                           hipStream_t st) {
    if (!c || !c->p2p_open || !dst || N > n_pad) return HEPT_ERR_ARG;
    const size_t bytes = (size_t)n_pad * D * 4, valid = (size_t)N * D * 4;
    if (bytes % 16 != 0 || valid % 16 != 0) return HEPT_ERR_SHAPE;
    const size_t blocks = (bytes / 16 + 255) / 256;
    // own slice: rows [rank * per, min(N, (rank + 1) * per)) -- per * D * 4 bytes is a multiple of 16 (checked by the combine)
    const size_t per = (size_t)n_pad / c->world, slice = per * D * 4;
    size_t lo = 0, hi = 0;
    if (skip_own && slice % 16 == 0) {
        lo = c->rank * slice / 16;
        hi = (c->rank + 1) * slice / 16;
        if (hi > valid / 16) hi = valid / 16;
        if (lo > hi) lo = hi;
    }
    // One workgroup per CU: every workgroup polls the same flag words, and reads of one uncached address are served
    // one after the other (1024 workgroups: ~6 us of polling before the first byte moved); 256 x 256 threads x 4 pieces
    // in flight are 4 MB of outstanding reads, more than the copy needs.
    static const size_t max_wgs = [] { const char* e = getenv("HEPT_COPY_OUT_WGS"); return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)HEPT_COPY_OUT_WGS; }();
    const unsigned grid = (unsigned)(blocks < max_wgs ? (blocks ? blocks : 1) : max_wgs);
    hipLaunchKernelGGL(wait_copy_out_kernel, dim3(grid), dim3(256), 0,
                       st, c->p2p_local, c->world, c->epoch, lay.out_off, bytes, valid, dst, c->d_state + HEPT_STATE_STATUS,
                       c->timeout_ticks, lo, hi, producer_signal() ? (char* const*)nullptr : (char* const*)c->d_peer, c->rank);
    return hept_launch_status();
}
