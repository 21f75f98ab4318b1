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
    unsigned int* counter;       // completion counter of the kernel that signals
    unsigned int* status;        // sticky error word
    unsigned long long timeout;  // bound of a wait, in wall_clock64 ticks
    int wait_groups;             // flags to wait for before reading received rows: wait_groups * world
    size_t slice_off;            // bytes from a buffer's start to this rank's slice of the gathered output
};

namespace {

constexpr int OUT_FLAG_WORD = HEPT_P2P_OUT_FLAG_WORD;

__device__ __forceinline__ unsigned int* flag_word(char* base, int idx) {
    return reinterpret_cast<unsigned int*>(base) + idx;
}

// 16 bytes to a (possibly remote) exchange buffer as two system-scope stores (global_store_dwordx2 sc0 sc1): written
// through to the destination whatever caching the mapping of a peer's buffer has, and counted by vmcnt until the
// write has been acknowledged -- which is what lets one fence per kernel (signal_when_all_done) stand for all of them.
__device__ __forceinline__ void store16_system(void* dst, const u32x4& v) {
    // (the pointer comes out of a table of peer bases: tell the compiler it is global memory, not a FLAT address)
    typedef unsigned long long __attribute__((address_space(1))) * gptr_t;
    gptr_t p = (gptr_t)(reinterpret_cast<unsigned long long*>(dst));
    __hip_atomic_store(p, ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, ((unsigned long long)v[3] << 32) | v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// true when the flag has reached `epoch` (wrap-safe); false on timeout (status bit set)
__device__ __forceinline__ bool wait_flag(const unsigned int* flag, unsigned int epoch, unsigned int* status, unsigned int bit,
                                          unsigned long long timeout) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;  // sticky
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) {
            atomicOr(status, bit);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: what the flag announces is read after it
    return true;
}

// The last workgroup of a kernel to get here raises `flag_idx` = epoch in every rank's buffer.
// Ordering: every store into an exchange buffer is a system-scope store (store16_system): written through, and the
// wave's vmcnt reaches zero only when it has been acknowledged.  __syncthreads() makes every wave of the workgroup wait
// for exactly that (workgroup-scope release = s_waitcnt vmcnt(0), no cache maintenance); the workgroup then counts
// itself in.  Only the single thread that sees the count complete pays for a system-scope fence before it writes the
// flags.  (A system-scope fence in every thread was the first build: thousands of L2 write-backs per launch while the
// block attention keeps the L2 dirty -- 153 us for a kernel that moves 12 us of data.)
__device__ __forceinline__ void signal_when_all_done(unsigned int* counter, char* const* peers, int world, int flag_idx,
                                                     unsigned int epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int s = 0; s < world; ++s)
                __hip_atomic_store(flag_word(peers[s], flag_idx), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


}  // namespace
