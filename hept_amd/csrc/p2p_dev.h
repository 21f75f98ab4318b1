// Device-side helpers of the one-sided exchange (p2p.hip; also the fused combine of combine.hip).
#pragma once
#include "common.h"

#define HEPT_P2P_OUT_FLAG_WORD (2048 / 4)
#define HEPT_MAX_RANKS_DEV 16   // = HEPT_MAX_RANKS of comm.h: pitch of the row-flag table [head group][source rank]

// What a kernel needs to take part in the one-sided exchange of one forward call.
struct P2pDev {
    char* const* peers;          // every rank's exchange buffer (device array; own rank: the local buffer)
    char* local;                 // this rank's exchange buffer
    int world, me;
    unsigned int epoch;
    unsigned int* status;        // sticky error word
    unsigned long long timeout;  // bound of a wait, in wall_clock64 ticks
    int wait_groups;             // flags to wait for before reading received rows: wait_groups * world
    size_t slice_off;            // bytes from a buffer's start to this rank's slice of the gathered output
    // HEPT_P2P_PRODUCER_SIGNAL=1 (A/B and fallback, see signal_when_all_done): the producers raise the flags themselves;
    // the kernels that wait must then NOT raise them at their start.  out_counter: the combine's completion counter.
    int consumer_raises;         // 1: the default protocol (a kernel boundary is the completion signal)
    unsigned int* out_counter;   // non-null: the fused combine counts its workgroups in and raises the output flag itself
    const float* self_rows;      // the rows this rank sent to itself: ordinary device memory, laid out like the receive region
    char* out_local;             // the caller's (n_pad, D) output, or null (view mode): this rank's own slice goes there
    size_t out_base;             // bytes from a buffer's start to row 0 of the gathered output
};

// One head group's table sum + one-sided push (p2p.hip: reduce_push_kernel), as a body that can also ride in the
// block-attention launch of the NEXT head group (block_attn.hip): the pushing workgroups mostly wait on the xGMI links,
// the attention workgroups on HBM gathers -- one launch overlaps the two on one stream, with no second stream and no
// event between them (a cross-stream hand-over costs ~7 us of bubble on the forking stream and ~15 us to the other).
struct PushArgs {
    const float* part;        // (Tl, N, H, row) per-table partial rows
    int Tl, N, H, h0, hg;     // heads [h0, h0 + hg) are summed and sent
    int per, world, me;
    char* const* peers;
    size_t recv_off, group_off;
    unsigned int epoch;
    int flag_idx;
    int push_wgs;             // workgroups of the launch that push (the first ones of the grid); 0: none
    // direct != 0 (one local table per rank, BASELINE config 4): nothing is summed and nothing is carried -- the block
    // attention of heads [h0, h0 + hg) stores every finished row straight into the receive buffer of the rank that
    // owns the point (direct_row below; `part`, `Tl`, `push_wgs` are unused).  No extra pass over the rows, no
    // separate push: only the drain of the last stores is exposed.
    int direct;
    // non-null (HEPT_P2P_PRODUCER_SIGNAL=1): the producing workgroups count themselves in here and the last one raises
    // flag_idx = epoch in every rank's buffer after a system-scope fence (the protocol of rounds 2-4)
    unsigned int* counter;
    // Rows whose owner is THIS rank never leave the GPU: they go to an ordinary (cached) mirror of the exchange buffer
    // (same offsets) instead of the uncached buffer the peers store into -- the combine reads its own "table" there.
    char* self;
};

namespace {

constexpr int OUT_FLAG_WORD = HEPT_P2P_OUT_FLAG_WORD;

__device__ __forceinline__ unsigned int* flag_word(char* base, int idx) {
    return reinterpret_cast<unsigned int*>(base) + idx;
}

// 16 bytes to a (possibly remote) exchange buffer as two system-scope stores (global_store_dwordx2 sc0 sc1): written
// through to the destination whatever caching the mapping of a peer's buffer has, and counted by vmcnt until the
// write has been acknowledged -- a wave cannot end before that, which is what raise_flags relies on.
__device__ __forceinline__ void store16_system(void* dst, const u32x4& v) {
    // (the pointer comes out of a table of peer bases: tell the compiler it is global memory, not a FLAT address)
    typedef unsigned long long __attribute__((address_space(1))) * gptr_t;
    gptr_t p = (gptr_t)(reinterpret_cast<unsigned long long*>(dst));
    __hip_atomic_store(p, ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, ((unsigned long long)v[3] << 32) | v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The communicator's device words (hept_comm::d_state): `status` points at word HEPT_STATE_STATUS; the two words at
// status + 2 hold the device address of a host-mapped copy of the status (hept_comm::h_status), written on the failure
// path only, so that the host sees a timeout at its next call without synchronising the device.
#define HEPT_STATE_STATUS 16
#define HEPT_STATE_HOSTPTR 18
__device__ __forceinline__ void record_timeout(unsigned int* status, unsigned int bit) {
    atomicOr(status, bit);
    unsigned int* host = reinterpret_cast<unsigned int*>(
        (unsigned long long)status[HEPT_STATE_HOSTPTR - HEPT_STATE_STATUS] |
        ((unsigned long long)status[HEPT_STATE_HOSTPTR - HEPT_STATE_STATUS + 1] << 32));
    // (a plain system-scope store, not an atomic OR: read-modify-write atomics on host memory need PCIe atomics; the
    //  host only asks "non-zero?", the exact bits are read from the device word by hept_comm_status)
    if (host) __hip_atomic_store(host, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool status_bad(const unsigned int* status) {
    return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// true when the flag has reached `epoch` (wrap-safe); false on timeout (status bit set, sticky: every later wait
// returns false at once, and every consumer of a wait's result poisons what it produces -- see wait_copy_out_kernel)
__device__ __forceinline__ bool wait_flag(const unsigned int* flag, unsigned int epoch, unsigned int* status, unsigned int bit,
                                          unsigned long long timeout) {
    if (status_bad(status)) return false;  // sticky
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) {
            record_timeout(status, bit);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: what the flag announces is read after it
    return true;
}

// Arrival flags are raised by the NEXT kernel of the stream, not by the kernel that stored the data: a kernel starts
// only after every wave of its predecessor has ended, and a wave ends only when its stores -- system-scope, written
// through (store16_system) -- have been acknowledged, so "the producer has finished" is known for free at the
// consumer's first instruction.  (Rounds 2-4 had the producers count themselves in: every workgroup drained its stores,
// met at a barrier and waited for a device-scope atomic to come back before it could retire -- which kept the block
// attention's workgroups resident ~45 % longer: 57 instead of 33 us for the one-table launch at tracking-60k.)
// Threads [0, n_flags * world) of the calling workgroup store flag[base + f * pitch] = epoch in every rank's buffer;
// the stores are idempotent, so any number of workgroups may do it (the callers use the first few of the grid: a
// workgroup that waits for this rank's own flag must never depend on a later workgroup getting a slot).
constexpr int RAISE_WGS = 16;
// Last statement of a kernel that stored into other ranks' buffers: the wave does not end before its stores have been
// acknowledged.  (The hardware waits for a wave's outstanding memory operations before it retires the wave anyway --
// S_ENDPGM implies S_WAITCNT 0 -- and the queue's end-of-kernel release waits for the dispatch's writes; this line
// makes the protocol's one assumption explicit in the code instead of leaving it to those two.)
__device__ __forceinline__ void drain_remote_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raise_flags(char* const* peers, int world, int n_flags, int base, int pitch,
                                            unsigned int epoch) {
    const int i = threadIdx.x;
    if (i < n_flags * world) {
        const int f = i / world, s = i - f * world;
        __hip_atomic_store(flag_word(peers[s], base + f * pitch), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// The protocol of rounds 2-4, kept as an A/B and a fallback (HEPT_P2P_PRODUCER_SIGNAL=1): nothing is inferred
// from a kernel boundary.  Every producing wave waits for the acknowledgement of its own stores, the workgroup meets at
// a barrier and counts itself in with a device-scope atomic; the workgroup that completes the count resets the counter,
// issues a system-scope fence and raises `flag_idx` = epoch in every rank's buffer.  Costs the producers their early
// retirement (one-table block attention 33 -> 57 us on the proxy), which is why it is not the default.
__device__ __forceinline__ void signal_when_all_done(unsigned int* counter, char* const* peers, int world, int flag_idx,
                                                     unsigned int epoch, unsigned int n_workgroups) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == n_workgroups - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int s = 0; s < world; ++s)
                __hip_atomic_store(flag_word(peers[s], flag_idx), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// direct mode: address of the partial row of point n, head h0 + hl, in the owner's receive buffer (same slot as
// reduce_push_body's: recv[g][me][n - dest * per][hl]); `remote` = the owner is another rank (system-scope stores)
__device__ __forceinline__ char* direct_row(const PushArgs& a, int n, int hl, int rowb, bool& remote) {
    const int dest = n / a.per;
    remote = dest != a.me;
    return (remote ? a.peers[dest] : a.self) + a.recv_off + a.group_off + (((size_t)a.me * a.per + (n - dest * a.per)) * a.hg + hl) * rowb;
}
__device__ __forceinline__ void store4_system(void* dst, unsigned int v) {
    typedef unsigned int __attribute__((address_space(1))) * gptr_t;
    __hip_atomic_store((gptr_t)(reinterpret_cast<unsigned int*>(dst)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Rows of heads [h0, h0 + hg) summed over the local tables; the row of point n goes to rank s = n / per, slot
// recv[g][me][n - s * per][h - h0].  One lane per 16-B piece of an output row (4 pieces per packed 64-B row, 8 per
// f32 row), consecutive lanes = consecutive pieces: a store instruction covers whole 64-B lines, which is what both
// uncached local memory and the xGMI links want.  Same arithmetic and row formats as reduce_heads_kernel: packed rows
// are widened to f32, summed in table order and rounded back.
template <bool P16>
__device__ __forceinline__ void reduce_push_body(const PushArgs& a, int wg_index) {
    constexpr int PIECES = P16 ? 4 : 8;        // 16-B pieces per row
    constexpr int ROWB = PIECES * 16;
    const unsigned int total = (unsigned int)a.N * a.hg * PIECES;   // < 2^31 (N * H * 8 pieces)
    const size_t tstride = (size_t)a.N * a.H * ROWB;  // bytes between tables
    const char* pbytes = reinterpret_cast<const char*>(a.part);
    const int Tl = a.Tl;
    for (unsigned int i = wg_index * blockDim.x + threadIdx.x; i < total; i += a.push_wgs * blockDim.x) {
        const unsigned int orow = i / PIECES;
        const int pc = (int)(i - orow * PIECES);
        const int n = (int)(orow / (unsigned int)a.hg);
        const int hl = (int)(orow - (unsigned int)n * a.hg);
        const char* src = pbytes + ((size_t)n * a.H + a.h0 + hl) * ROWB + pc * 16;
        u32x4 v = *reinterpret_cast<const u32x4*>(src);
        if (Tl > 1) {
            u32x4 x[2];
            const int tpre = Tl - 1 < 2 ? Tl - 1 : 2;   // the usual three tables: all loads in flight at once
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (t < tpre) x[t] = *reinterpret_cast<const u32x4*>(src + (size_t)(t + 1) * tstride);
            if (P16 && pc < 3) {  // 8 bf16 numerators
                float s[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[2 * j] = hept_bf16_lo(v[j]); s[2 * j + 1] = hept_bf16_hi(v[j]); }
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (t < tpre) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s[2 * j] += hept_bf16_lo(x[t][j]); s[2 * j + 1] += hept_bf16_hi(x[t][j]); }
                    }
                for (int t = 3; t < Tl; ++t) {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(src + (size_t)t * tstride);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s[2 * j] += hept_bf16_lo(y[j]); s[2 * j + 1] += hept_bf16_hi(y[j]); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = hept_pack_bf16(s[2 * j], s[2 * j + 1]);
            } else {              // f32 words (a packed row's last piece: [denominator, 0, 0, 0])
                float s[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (t < tpre) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] += __uint_as_float(x[t][j]);
                    }
                for (int t = 3; t < Tl; ++t) {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(src + (size_t)t * tstride);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += __uint_as_float(y[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(s[j]);
                if (P16) v[1] = v[2] = v[3] = 0u;
            }
        }
        const int dest = n / a.per;
        char* row = (dest == a.me ? a.self : a.peers[dest]) + a.recv_off + a.group_off +
                    (((size_t)a.me * a.per + (n - dest * a.per)) * a.hg + hl) * ROWB;
        // own rows: ordinary memory, one 16-B store
        if (dest == a.me) *reinterpret_cast<u32x4*>(row + pc * 16) = v;
        else store16_system(row + pc * 16, v);
    }
    // (no signal: the flags of every head group are raised by the kernel that waits for the rows, see raise_flags --
    //  unless the producers have been asked to signal themselves)
    drain_remote_stores();
    if (a.counter) signal_when_all_done(a.counter, a.peers, a.world, a.flag_idx, a.epoch, (unsigned int)a.push_wgs);
}

}  // namespace
