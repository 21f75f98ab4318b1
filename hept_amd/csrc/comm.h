// Communicator of the table-sharded operator: one process per GPU, one exchange step over xGMI.
// Two transports behind one object:
//   * RCCL (comm.hip): ncclAllToAll / ncclAllGather, bound at run time (dlopen of librccl.so.1 -- inside a PyTorch
//     process that resolves to the copy torch has already loaded), so libhept_hip.so has no link-time dependency on it
//     and loads in a plain C host;
//   * one-sided stores (p2p.hip): every rank maps the other ranks' exchange buffers (HIP IPC handles of uncached
//     device memory) and the kernels that produce a rank's rows store them straight into the owner's buffer; arrival
//     is signalled with epoch flags.  No collective launch on the critical path.
#pragma once
#include "common.h"

#define HEPT_MAX_RANKS 16   // (p2p_dev.h: HEPT_MAX_RANKS_DEV)
// exchange buffer, first HEPT_P2P_FLAG_BYTES: [HEPT_MAX_HEAD_GROUPS][HEPT_MAX_RANKS] u32 row flags at 0, [HEPT_MAX_RANKS] output flags at 2048

struct hept_comm {
    void* nccl = nullptr;        // ncclComm_t, or null for a communicator without RCCL (one-sided transport only)
    int rank = 0, world = 1, device = 0;
    hipStream_t side = nullptr;  // transfers of finished head groups run here, behind the block attention
    hipEvent_t fork[HEPT_MAX_HEAD_GROUPS] = {};
    hipEvent_t join = nullptr;
    // ---- one-sided transport
    char* p2p_local = nullptr;   // this rank's exchange buffer: [flags | received rows | gathered output]
    char* p2p_self = nullptr;    // ordinary device memory of the same size and layout: the rows this rank sends to itself
    size_t p2p_bytes = 0;
    char* p2p_peer[HEPT_MAX_RANKS] = {};  // every rank's buffer as mapped here (own rank: p2p_local)
    bool p2p_open = false;
    char** d_peer = nullptr;     // device copy of p2p_peer
    unsigned int* d_state = nullptr;  // device words: [0], [1] completion counters of the row / output producers (used with
                                      // HEPT_P2P_PRODUCER_SIGNAL=1 only: p2p.hip producer_signal), [16] status,
                                      // [18, 19] device address of h_status (p2p_dev.h: HEPT_STATE_*)
    unsigned int* h_status = nullptr; // host-mapped copy of the status word: a kernel whose wait timed out writes it
    bool broken = false;              // a step failed on the host after it had taken its epoch: the ranks' epochs may
                                      // differ, the transport refuses further steps until hept_comm_reset_status
    unsigned int epoch = 0;      // one per forward call, the same on every rank
    unsigned long long timeout_ticks = 0;  // bound of a device-side wait in wall_clock64 ticks (HEPT_P2P_TIMEOUT_S, 20 s)
    // Output gather without the copy (hept_comm_set_out_view): the gathered output stays in the exchange buffer, in one
    // of two regions taken in turn (epoch parity), and hept_comm_out_view hands the caller the region of the last step
    bool out_view = false;
    const float* last_out = nullptr;
};

// all-to-all of `bytes_per_peer` bytes per rank pair on `st` (ncclAllToAll on bytes)
int hept_comm_all_to_all(hept_comm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t st);
// in-place all-gather: rank r's `count` floats already sit at buf + r * count
int hept_comm_all_gather_f32(hept_comm* c, float* buf, size_t count, hipStream_t st);
void hept_comm_set_error(const char* what, const char* detail);
int hept_comm_init_streams(hept_comm* c);   // side stream + events (shared by both constructors)
void hept_p2p_release(hept_comm* c);        // unmap / free the one-sided buffers

// one-sided transport (p2p.hip); layouts in bytes from the start of a rank's exchange buffer
struct P2pLayout {
    size_t recv_off, out_off, out_bytes, bytes;   // two gathered-output regions of out_bytes each at out_off
};
P2pLayout hept_p2p_layout(int N, int H, int D, int world, int precision);
// sum over the local tables of heads [h0, h0 + hg) -> rows stored into the owners' receive buffers, then flags
// (mirror: this rank's own rows go to hept_comm::p2p_self instead of its uncached buffer -- the fused combine reads them there)
int hept_p2p_reduce_push(hept_comm* c, const float* part, int part_precision, int Tl, int N, int H, int D, int h0,
                         int hg, int g, int acc_precision, const P2pLayout& lay, bool mirror, hipStream_t st);
// the same work described as kernel arguments, for a block-attention launch that carries it as extra workgroups
struct PushArgs;
int hept_p2p_push_args(hept_comm* c, const float* part, int part_precision, int Tl, int N, int H, int D, int h0, int hg,
                       int g, int acc_precision, const P2pLayout& lay, int push_wgs, bool mirror, PushArgs* out);
// one local table per rank: arguments of a block-attention launch that scatters its rows straight into the owners'
// receive buffers and raises the flags itself (PushArgs::direct)
int hept_p2p_direct_args(hept_comm* c, int N, int H, int D, int h0, int hg, int g, int acc_precision,
                         const P2pLayout& lay, bool mirror, PushArgs* out);
int hept_block_attn_heads_push(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos, int N,
                               int H, int D, int Tl, int B, int precision, int h0, int hg, int hout, int hsub,
                               int n_rows_out, float* part, const PushArgs* push, void* stream,
                               VSrc vs = VSrc{});   // block_attn.hip (vs: value rows read from the caller's v, f32 rows only)
int hept_p2p_wait_rows(hept_comm* c, int head_groups, hipStream_t st);
// this rank's finished (cnt, D) rows (already in its slice of the local output region) -> every other rank, then flags
int hept_p2p_push_out(hept_comm* c, int per, int D, const P2pLayout& lay, hipStream_t st);
// the three steps above fused into the combine (D == 24): wait for the rows, combine `cnt` >= 1 points of this rank,
// store them into every rank's gathered output, raise the output flag
// (own rows are read in p2p_self; out_local: the caller's (n_pad, D) output or null -- own slice is stored there directly)
int hept_p2p_combine_push(hept_comm* c, int head_groups, int per, int cnt, int H, int hg, int acc_precision,
                          const float* out_weight, const float* out_bias, const P2pLayout& lay, float* out_local,
                          hipStream_t st);
// wait for every rank's slice, then copy the gathered (n_pad, D) output to `dst`
// (rows [N, n_pad) are written as zeros; the whole output as NaN when a wait of the step timed out)
// (skip_own: this rank's slice has been stored into dst by the combine itself)
int hept_p2p_wait_copy_out(hept_comm* c, int n_pad, int N, int D, const P2pLayout& lay, float* dst, bool skip_own,
                           hipStream_t st);
// view mode: wait for every rank's slice; the gathered output stays where it is (NaN-filled when a wait timed out)
int hept_p2p_wait_out(hept_comm* c, int N, int D, const P2pLayout& lay, hipStream_t st);
int hept_p2p_failed(const hept_comm* c);   // non-zero: a wait timed out or a step failed mid-way (no device sync)
