/*
 * hept_hip.h — C ABI of the MI355X (gfx950) HEPT block-attention hot path.
 *
 * Drop-in boundary for the reference operator HEPTAttention.forward
 * (reference example/hept.py:43-81, helpers example/hept_utils.py:38-97).  The
 * reference is pure Python/eager PyTorch and has no FFI of its own; each entry
 * point below replaces the sequence of ATen ops named in its comment.  Plain
 * device pointers and sizes only — no torch types.  All functions are
 * asynchronous on `stream` (a hipStream_t passed as void*), never allocate,
 * never synchronise, and return 0 on success or a HEPT_ERR_* code.
 *
 * Notation: N points (multiple of block size B), H heads, D head dim,
 * C coordinate dim, E = D + C hash dim, T tables ("n_hashes"), Tl tables
 * handled by this call (table sharding: tables [t0, t0+Tl) of T).
 *
 * Device layouts (all row-major, contiguous):
 *   q,k,v        (N, H*D)   f32      inputs
 *   coords       (N, C)     f32
 *   w_rpe        (H*D, (C-1)*K) f32  the nn.Linear weight of the caller's w_rpe
 *   alpha        (H, E, T)  f32      E2LSH projection
 *   codes        (T, H, N)  i64      combined_shifts (AND code)
 *   sqrt_w       (H, C)     f32      sqrt(2*sum_k exp(min(sum_d w,50)))
 *   qhat         (H, N, 32) tile     augmented query rows, see DESIGN.md §3
 *   kvhat        (H, N, 64) tile     augmented key row | value row
 *   qproj,kproj  (Tl, H, N) f32      real-valued E2LSH hashes
 *   qpos,kpos    (Tl, H, N) i32      ascending stable sort permutations
 *   part         (Tl, N, H, row)     per-table partial rows, two formats:
 *                                      HEPT_PREC_F32 : 32 f32 = [numer(0..D-1) | denom(D) | 0]        (128 B)
 *                                      HEPT_PREC_BF16: 16 dwords = [24 bf16 numer | f32 denom | 0]    ( 64 B),
 *                                      written by hept_block_attn iff the tiles are 16-bit and D == 24
 *   acc          (N, H, row)         sum over tables of part: f32 rows (N, H, 32), or packed rows again when the
 *                                      caller asks for acc_precision HEPT_PREC_BF16 (table-sharding exchange)
 *   out          (N, D)     f32
 * "tile" element type is f32 (precision 0) or 16-bit (precision 1: bf16; precision 2: fp16 for q^/k^ rows, bf16 for v rows).
 */
#ifndef HEPT_HIP_H
#define HEPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEPT_OK 0
#define HEPT_ERR_SHAPE 1   /* unsupported or inconsistent sizes */
#define HEPT_ERR_LAUNCH 2  /* HIP reported a launch error */
#define HEPT_ERR_ARG 3     /* null pointer / workspace too small */
#define HEPT_ERR_COMM 4    /* RCCL unavailable or reported an error: see hept_comm_last_error() */

#define HEPT_PREC_F32 0      /* f32 tiles: the reference's numerics.  The tile products run on the bf16 matrix pipe
                                with every f32 factor split into bf16 pieces (6 products for the logits, 3 for
                                P.V), which reproduces the f32 products to f32 accuracy at ~1/3 of the cycles */
#define HEPT_PREC_BF16 1     /* bf16 tiles (q^, k^, v, weights P), bf16 MFMA, packed bf16 partial numerators */
#define HEPT_PREC_MIXED16 2  /* as BF16 but q^/k^ tiles in fp16 (11-bit significand protects the logit
                                q.k - |q|^2/2 - |k|^2/2; values are clamped to +-65504); P, v stay bf16 */

#define HEPT_PREC_F32_MFMA 3 /* f32 tiles on v_mfma_f32_32x32x2_f32 (the exact f32 fma chain); same storage and row
                                formats as HEPT_PREC_F32, ~2x slower -- kept as the in-library ground truth */

#define HEPT_PREC_F32_DIFF 4 /* as HEPT_PREC_F32_MFMA, but the coordinate part of the logit (every row column from D on) is
                                formed as -(q^_c - k^_c)^2 / 2 from the difference of the stored values instead of
                                q^.k^ - |q^|^2/2 - |k^|^2/2: no cancellation when sqrt_w . coords is large (a trained
                                w_rpe on un-normalised coordinates: terms of ~3e8 whose f32 sum is noise).  The mode for
                                such inputs; mathematically the same operator (example/hept.py:8-12) */

#define HEPT_ROW 32          /* padded row width (elements) of qhat / k / v / part rows */
#define HEPT_MAX_TABLES 8    /* tables per hept_prep_hash / hept_sort_tables call; the whole-operator entry points
                                take any number of tables and walk them in chunks of this size */
#define HEPT_MAX_BLOCK 256   /* largest block_size */

/* ABI version of this library (bumped on any signature change). */
int hept_abi_version(void);

/* 0 if (N,H,D,C,T_local,B) is supported by the kernels, else HEPT_ERR_SHAPE. */
int hept_check_shape(int N, int H, int D, int C, int Tl, int B);

/* Bytes of scratch hept_forward / hept_forward_partial need. */
size_t hept_workspace_bytes(int N, int H, int D, int C, int Tl, int B, int precision);

/* replaces prep_qk's weight math, example/hept.py:22-23,25 (and the rearrange at :48-54) */
int hept_rpe_scale(const float* w_rpe, int H, int D, int C, int K, float* sqrt_w, void* stream);

/* replaces prep_qk (example/hept.py:25-27), the head-major rearranges (:57-59), E2LSH.forward
 * (example/hept_utils.py:45-47) and the min/max of lsh_mapping (:66-70).
 * minmax: (Tl, H, HEPT_PREP_GRID, 4) f32 per-workgroup partials [hash min, hash max, largest AND
 * code seen, 0]; reduced by hept_sort_tables.  The code maximum only scales the sort's bucket ids (the
 * sort is exact for any value), so the tuned kernels take it from a fixed sample of the points. */
/* raw_size < N selects the padding rule of the reference's src variant (src/models/attention/hept.py:89-96):
 * rows >= raw_size are zero rows that hash to +inf (raw_size == N: no such rows).  codes may be NULL (src
 * variant: the key range then comes from hept_sort_tables_src). */
int hept_prep_hash(const float* q, const float* k, const float* v, const float* coords,
                   const float* sqrt_w, const float* alpha, const int64_t* codes,
                   int N, int raw_size, int H, int D, int C, int T, int t0, int Tl, int precision,
                   void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax,
                   void* stream);
#define HEPT_PREP_GRID 1024

/* replaces hash_shift (example/hept_utils.py:70), the AND-shift add (example/hept.py:63-65)
 * and both argsorts (:67-68) with an exact stable sort of the fp32 keys (bucket pass + in-bucket ranking).
 * sort_ws: hept_sort_workspace_bytes(N, H, Tl) bytes. */
size_t hept_sort_workspace_bytes(int N, int H, int Tl);
int hept_sort_tables(const float* qproj, const float* kproj, const int64_t* codes,
                     const float* minmax, int N, int H, int T, int t0, int Tl,
                     void* sort_ws, int32_t* qpos, int32_t* kpos, void* stream);

/* The reference's src variant: keys = hash + get_geo_shift (src/models/attention/hept.py:46-56,
 * 98-101): shift = (phi_idx * span) * cfac + eta_idx * span with every product / sum rounded separately, where
 * eta_idx, phi_idx (T, H, N) f32 are the caller's region_indices and cfac (T, H) f32 = ceil(regions_h[0]) + 1. */
int hept_sort_tables_src(const float* qproj, const float* kproj, const float* eta_idx, const float* phi_idx,
                         const float* cfac, float* minmax, int N, int H, int T, int t0, int Tl,
                         void* sort_ws, int32_t* qpos, int32_t* kpos, void* stream);

/* Generic form of the same sort: stable ascending argsort of S segments of L fp32 keys each
 * (row-major (S, L); +inf is a legal padding key and sorts last).  pos (S, L) i32.  Used by
 * hept_prepare_input; equals torch.sort(stable=True).indices. */
size_t hept_argsort_workspace_bytes(int S, int L);
int hept_segmented_argsort(const float* keys, int S, int L, void* ws, int32_t* pos, void* stream);
/* Ragged form: only the first seg_len[s] (device i32, 0 <= seg_len[s] <= L) keys of segment s take part; pos[s][0 ..
 * seg_len[s]) is their stable ascending permutation, the rest of the row is left untouched.  (Per-cloud sorts of a
 * batch of clouds of different sizes: padding every segment with +inf keys would sort the padding too.) */
int hept_segmented_argsort_ragged(const float* keys, int S, int L, const int32_t* seg_len, void* ws,
                                  int32_t* pos, void* stream);

/* replaces sort_to_buckets x3 (example/hept.py:70-72), qkv_res (:7-18), invert_permutation and
 * unsort_from_buckets x2 (:76-78): gather -> block-local RBF attention on MFMA -> scatter. */
int hept_block_attn(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos,
                    int N, int H, int D, int Tl, int B, int precision, float* part, void* stream);

/* The same for heads [h0, h0 + hg) only (table sharding sends head groups one at a time): the row
 * of (table t, point n, head h) goes to row index (t * n_rows_out + n) * hout + (h - hsub) of `part`
 * (hept_block_attn = h0 0, hg H, hout H, hsub 0, n_rows_out N).  n_rows_out >= N. */
int hept_block_attn_heads(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos,
                          int N, int H, int D, int Tl, int B, int precision, int h0, int hg, int hout, int hsub,
                          int n_rows_out, float* part, void* stream);

/* Format of the partial rows hept_block_attn writes for (precision, D): HEPT_PREC_BF16 (packed) or
 * HEPT_PREC_F32. */
int hept_part_precision(int precision, int D);

/* acc = sum_t part[t] (table-sharded ranks exchange `acc` between GPUs afterwards).  acc_precision
 * HEPT_PREC_F32: f32 rows (N, H, 32).  HEPT_PREC_BF16 (packed input only): packed 64-B rows again
 * (numerators rounded to bf16 once more, denominators exact) -- half the bytes on the xGMI links. */
int hept_reduce_tables(const float* part, int part_precision, int Tl, int N, int H, int D, float* acc,
                       int acc_precision, void* stream);

/* One head group of the same sum: dst (n_pad, hg, row) = sum_t part[t][n][h0 + j] for j < hg; points [N, n_pad)
 * (padding up to a multiple of the rank count) get zero rows. */
int hept_reduce_heads(const float* part, int part_precision, int Tl, int N, int H, int D, int h0, int hg,
                      int n_pad, float* dst, int acc_precision, void* stream);

/* replaces the cross-table combine (example/hept.py:79) and out_linear (:80) for points
 * [n0, n0+n_count): out[n] = bias + W . (sum_t numer / sum_t denom).  `part` may hold Tl >= 1
 * tables (Tl == 1: an already reduced `acc`).  out points at row n0 of the (N, D) output.
 * D == 24: `part` and `out_weight` must be 16-byte aligned (rows and weight columns are read as 16-B pieces);
 * HEPT_ERR_ARG otherwise. */
int hept_combine_out(const float* part, int part_precision, int Tl, int N, int H, int D, int n0,
                     int n_count, const float* out_weight, const float* out_bias, float* out,
                     void* stream);

/* hept_combine_out on partial rows that arrive split by head groups: the rows of heads [g*HG, (g+1)*HG) live in
 * their own (Tl, N, HG, row) buffer at part + g * group_stride (in 4-byte units); H % HG == 0.  HG == H is
 * hept_combine_out.  (Table sharding: the "tables" are the slices received from the ranks, one buffer per
 * exchanged head group.) */
int hept_combine_groups(const float* part, int part_precision, int Tl, int N, int H, int D, int n0,
                        int n_count, int HG, size_t group_stride, const float* out_weight,
                        const float* out_bias, float* out, void* stream);

/* Whole operator for tables [t0, t0+Tl): everything above in one call.
 * K > 0: `w_rpe` is w_rpe.weight (H*D, (C-1)*K); its scale sqrt_w (H, C) (example/hept.py:22-25) is computed inside
 * the row builder's prologue on EVERY call -- nothing derived from a parameter is remembered between calls, so an
 * in-place update of the weight is always seen (the reference recomputes it every forward as well); H*(C-1)*K <= 1024.
 * K == 0 (every whole-operator entry point): `w_rpe` is the (H, C) result of hept_rpe_scale instead -- for a caller
 * that manages the scale itself.
 * hept_forward writes out (N, D); hept_forward_partial stops at acc (N, H, row) in the row format
 * acc_precision (HEPT_PREC_F32, or hept_part_precision(precision, D)). */
int hept_forward(const float* q, const float* k, const float* v, const float* coords,
                 const int64_t* codes, const float* w_rpe, const float* alpha,
                 const float* out_weight, const float* out_bias,
                 int N, int H, int D, int C, int K, int T, int B, int precision,
                 void* workspace, size_t workspace_bytes, float* out, void* stream);
int hept_forward_partial(const float* q, const float* k, const float* v, const float* coords,
                         const int64_t* codes, const float* w_rpe, const float* alpha,
                         int N, int H, int D, int C, int K, int T, int t0, int Tl, int B,
                         int precision, int acc_precision, void* workspace, size_t workspace_bytes,
                         float* acc, void* stream);

/* Table sharding with the exchange pipelined behind the block attention: hept_partial_begin runs
 * everything up to the sort for tables [t0, t0+Tl) and leaves rows and permutations in `workspace`
 * (hept_workspace_bytes(N,H,D,C,Tl,B,precision)); hept_partial_heads then runs the block attention for heads
 * [h0, h0+hg) and writes the sum over the local tables as dst (n_pad, hg, row) in the row format acc_precision --
 * the caller sends that head group to the other ranks while the next group is computed.  The same workspace and
 * sizes must be passed to both; the sequence begin, heads(0..), heads(..) is stream-ordered. */
int hept_partial_begin(const float* q, const float* k, const float* v, const float* coords,
                       const int64_t* codes, const float* w_rpe, const float* alpha,
                       int N, int H, int D, int C, int K, int T, int t0, int Tl, int B, int precision,
                       void* workspace, size_t workspace_bytes, void* stream);
int hept_partial_begin_src(const float* q, const float* k, const float* v, const float* coords,
                           const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                           const float* w_rpe, const float* alpha,
                           int N, int H, int D, int C, int K, int T, int t0, int Tl, int B, int precision,
                           void* workspace, size_t workspace_bytes, void* stream);
int hept_partial_heads(void* workspace, size_t workspace_bytes, int N, int H, int D, int C, int Tl, int B,
                       int precision, int h0, int hg, int n_pad, int acc_precision, float* dst, void* stream);

/* ---- Table sharding over the GPUs of one node, RCCL over xGMI -------------------------------------
 * The reference is single-process; its only cross-table coupling is `out = o.sum(0) / logits.sum(0)`
 * (example/hept.py:79).  One process per GPU owns tables [t0, t0+Tl) and holds a hept_comm: an RCCL communicator
 * (bound with dlopen at first use, so this library has no link-time dependency on RCCL), a side stream and the
 * events that order it against the caller's stream.  hept_forward_sharded runs the whole operator on replicated
 * inputs in ONE call:
 *   begin (rows, hashes, sort)  ->  for each of `head_groups` head groups: block attention + sum over the local
 *   tables into the send buffer, then ncclAllToAll of that group on the side stream (rank r receives points
 *   [r*per, (r+1)*per), per = ceil(N / world)) while the caller's stream computes the next group  ->  join  ->
 *   combine of the received slices + out_linear for this rank's points, written into its slice of
 *   out_full (world*per, D)  ->  in-place ncclAllGather of out_full on the caller's stream.
 * Rows travel in the format hept_part_precision(precision, D) (packed 64-B rows for 16-bit tiles with D == 24,
 * f32 rows otherwise).  xbuf: hept_exchange_bytes(...) bytes of device scratch that must stay untouched between
 * calls in flight; out_full rows [0, N) are the result (every rank ends with all of them).
 * Creating a communicator: rank 0 calls hept_comm_unique_id, the HEPT_COMM_ID_BYTES bytes reach every rank by any
 * host-side means, every rank (with its GPU current) calls hept_comm_create -- collective and blocking. */
#define HEPT_MAX_HEAD_GROUPS 8
#define HEPT_COMM_ID_BYTES 128
typedef struct hept_comm hept_comm;
int hept_comm_unique_id(void* id128);
int hept_comm_create(const void* id128, int rank, int world, hept_comm** out);
int hept_comm_destroy(hept_comm* comm);
int hept_comm_rank(const hept_comm* comm);
int hept_comm_world(const hept_comm* comm);
int hept_comm_has_rccl(const hept_comm* comm);
const char* hept_comm_last_error(void);
size_t hept_exchange_bytes(int N, int H, int D, int world, int precision);

/* Second transport of the same exchange: one-sided stores over xGMI, no collective launch on the critical path.
 * Every rank allocates an exchange buffer of hept_p2p_bytes(...) bytes of uncached device memory
 * (hept_comm_p2p_alloc), the HEPT_IPC_HANDLE_BYTES-byte HIP IPC handles reach every rank by any host-side means,
 * and every rank maps the others' buffers (hept_comm_p2p_open, handles in rank order).  With transport
 * HEPT_TRANSPORT_ONE_SIDED hept_forward_sharded then stores each table-summed row straight into the buffer of the
 * rank that finishes its point; the kernel that follows on the stream (the combine) raises the epoch flags that
 * announce them -- a kernel boundary says that every store of the kernels before it has been acknowledged -- and polls
 * its own flags before it reads; it stores its finished output slice into every rank's buffer, and the gather kernel
 * raises the output flag and copies the gathered output to out_full once every slice has arrived (xbuf is not used).
 * Calls are collective: every rank makes the same sequence.
 * One local table per rank (Tl == 1, BASELINE config 4): the block attention itself stores every finished row into
 * the owner's buffer (16-byte pieces of 64-byte rows) -- no table sum, no separate push.
 * A wait is bounded (20 s, HEPT_P2P_TIMEOUT_S; sticky).  After a timeout the step's output is written as NaN on the
 * rank that timed out and the slice it sends to the others is NaN as well -- a lost or slow peer never turns into
 * plausible numbers -- and every later hept_forward_sharded on that communicator returns HEPT_ERR_COMM at once (the
 * kernel that timed out wrote a host-mapped status word: no device synchronisation is needed to see it).
 * hept_comm_status reports bit 0 (rows) / bit 1 (output) / bit 2 (a step failed on the host after taking its epoch).
 * hept_comm_reset_status clears the failure AND restarts the protocol (epoch, flags, counters): every rank must call
 * it, with a host barrier before the next exchange -- or fall back to HEPT_TRANSPORT_RCCL.  hept_comm_create_local
 * makes a communicator without RCCL (one-sided transport only; at most 16 ranks). */
#define HEPT_TRANSPORT_RCCL 0
#define HEPT_TRANSPORT_ONE_SIDED 1
#define HEPT_IPC_HANDLE_BYTES 64
int hept_comm_create_local(int rank, int world, hept_comm** out);
size_t hept_p2p_bytes(int N, int H, int D, int world, int precision);
int hept_comm_p2p_alloc(hept_comm* comm, size_t bytes, void* handle_out);
int hept_comm_p2p_open(hept_comm* comm, const void* handles);
int hept_comm_p2p_ready(const hept_comm* comm, size_t bytes);
int hept_comm_status(hept_comm* comm, int* status);
int hept_comm_reset_status(hept_comm* comm);   /* collective restart after a failure (synchronises the device) */
/* debugging aid: this rank's HEPT_P2P_FLAG_BYTES bytes of arrival flags ([head group][source rank] u32 at byte 0,
 * output flags [source rank] u32 at byte 2048; each holds the epoch of the last arrival) and its own epoch */
#define HEPT_P2P_FLAG_BYTES 4096
int hept_comm_p2p_flags(hept_comm* comm, void* out_flags, unsigned int* epoch);
/* Output gather WITHOUT the final copy (one-sided transport; opt-in, collective: every rank sets the same mode between
 * steps).  With the mode on, hept_forward_sharded accepts out_full == NULL: the gathered (N, D) f32 output stays in this
 * rank's exchange buffer -- uncached device memory -- and hept_comm_out_view returns its address after the call (work
 * that reads it must be ordered after the call on `stream`).  The steps alternate between two output regions, so a
 * view stays intact during the NEXT hept_forward_sharded on this communicator and is overwritten by the one after it:
 * consume (or copy) it before the second next call.  After a timed-out wait the view holds NaN, like out_full would.
 * (What it saves is a 4 N D-byte copy out of uncached memory behind the last flag: 11 -> 3 us at tracking-60k.) */
int hept_comm_set_out_view(hept_comm* comm, int on);
int hept_comm_out_view(hept_comm* comm, const float** out);
int hept_forward_sharded(hept_comm* comm, const float* q, const float* k, const float* v, const float* coords,
                         const int64_t* codes, const float* w_rpe, const float* alpha,
                         const float* out_weight, const float* out_bias,
                         int N, int H, int D, int C, int K, int T, int t0, int Tl, int B, int precision,
                         int head_groups, int transport, void* workspace, size_t workspace_bytes, void* xbuf,
                         size_t xbuf_bytes, float* out_full, void* stream);
int hept_forward_sharded_src(hept_comm* comm, const float* q, const float* k, const float* v, const float* coords,
                             const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                             const float* w_rpe, const float* alpha, const float* out_weight, const float* out_bias,
                             int N, int H, int D, int C, int K, int T, int t0, int Tl, int B, int precision,
                             int head_groups, int transport, void* workspace, size_t workspace_bytes, void* xbuf,
                             size_t xbuf_bytes, float* out_full, void* stream);

/* SURVEY.md §8 f-3 — the reference's src variant of the same operator (src/models/attention/hept.py:74-117, caller
 * src/models/baselines/transformer.py:43-57): no AND codes; the sort key is hash + get_geo_shift (see
 * hept_sort_tables_src) and rows >= raw_size are padding (zeroed q^, k^, v; hash +inf).  eta_idx / phi_idx (T, H, N)
 * f32 = kwargs["region_indices"] viewed "(c h) n -> c h n"; cfac (T, H) f32 = ceil(kwargs["regions_h"][0]) + 1. */
int hept_forward_src(const float* q, const float* k, const float* v, const float* coords,
                     const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                     const float* w_rpe, const float* alpha, const float* out_weight, const float* out_bias,
                     int N, int H, int D, int C, int K, int T, int B, int precision,
                     void* workspace, size_t workspace_bytes, float* out, void* stream);
int hept_forward_partial_src(const float* q, const float* k, const float* v, const float* coords,
                             const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                             const float* w_rpe, const float* alpha,
                             int N, int H, int D, int C, int K, int T, int t0, int Tl, int B,
                             int precision, int acc_precision, void* workspace, size_t workspace_bytes,
                             float* acc, void* stream);

/* SURVEY.md §8 f-4 — the Attn block around the operator (example/transformer.py:131-165), eval mode, D == 24:
 *   x_normed = norm1(x); q,k,v = w_q/w_k/w_v(x_normed); aggr = HEPTAttention(q,k,v,...);
 *   x = x + aggr; y = x + ff(norm2(x))                                  (dropout is the identity in eval)
 * hept_prep_hash_fused = hept_prep_hash with LayerNorm and the three bias-free projections computed while the rows
 * are staged: x is (N, D), q/k/v never exist in HBM.  hept_combine_ffn = hept_combine_out with the residual,
 * norm2 and the two-layer feed-forward in the epilogue; x and y point at row n0.  hept_attn_block_forward runs
 * the whole block (rpe_scale, prep_hash_fused, sort_tables, block_attn, combine_ffn) in one call. */
typedef struct {
    const float* norm1_w;  /* (D) */
    const float* norm1_b;
    const float* w_q;      /* (H*D, D), no bias */
    const float* w_k;
    const float* w_v;
    const float* w_rpe;    /* (H*D, (C-1)*K) */
    const float* alpha;    /* (H, D+C, T) */
    const float* out_w;    /* (D, H*D) */
    const float* out_b;    /* (D) or NULL */
    const float* norm2_w;
    const float* norm2_b;
    const float* ff1_w;    /* ff.0: (D, D), (D) */
    const float* ff1_b;
    const float* ff2_w;    /* ff.2 */
    const float* ff2_b;
    float eps1, eps2;      /* LayerNorm eps of norm1 / norm2 */
} hept_attn_params;

int hept_prep_hash_fused(const float* x, const float* norm_w, const float* norm_b, float eps,
                         const float* w_q, const float* w_k, const float* w_v, const float* coords,
                         const float* sqrt_w, const float* alpha, const int64_t* codes,
                         int N, int raw_size, int H, int D, int C, int T, int t0, int Tl, int precision,
                         void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax, void* stream);
int hept_combine_ffn(const float* part, int part_precision, int Tl, int N, int H, int D, int n0, int n_count,
                     const float* out_weight, const float* out_bias, const float* x,
                     const float* norm_w, const float* norm_b, float eps,
                     const float* ff1_w, const float* ff1_b, const float* ff2_w, const float* ff2_b,
                     float* y, void* stream);
int hept_attn_block_forward(const float* x, const float* coords, const int64_t* codes,
                            const hept_attn_params* params, int N, int H, int D, int C, int K, int T, int B,
                            int precision, void* workspace, size_t workspace_bytes, float* y, void* stream);

/* SURVEY.md §8 f-2 — backward of the block attention (the reference trains through example/hept.py:55-80 with
 * plain autograd; there is no custom backward to mirror).  f32 tiles only; the tile products run as split-bf16
 * MFMAs (6 or 5 bf16 products per f32 product, see HEPT_PREC_F32).  gacc (N, H, 32) f32 is the
 * gradient of the table-summed partial rows [d numer | d den | 0]; qhat/kvhat/qpos/kpos are the forward's.
 * dq_part (Tl, N, H, 32) receives d q^ rows, dkv_part (Tl, N, H, 64) receives [d k^ | d v] rows (point order,
 * one row per table).  hept_bwd_reduce sums the tables and undoes the augmentation: dq, dk, dv (N, H*D) and
 * dcs (N, H, C) = gradient of the scaled coordinates sqrt_w[h,c] * coords[n,c]. */
int hept_block_attn_bwd(const float* qhat, const float* kvhat, const int32_t* qpos, const int32_t* kpos,
                        const float* gacc, int N, int H, int D, int Tl, int B, float* dq_part,
                        float* dkv_part, void* stream);
/* the same on v_mfma_f32_32x32x2_f32 (exact f32 fma chains, ~2.5x slower): the in-library ground truth of the
 * split-bf16 kernel behind hept_block_attn_bwd */
int hept_block_attn_bwd_f32mfma(const float* qhat, const float* kvhat, const int32_t* qpos, const int32_t* kpos,
                                const float* gacc, int N, int H, int D, int Tl, int B, float* dq_part,
                                float* dkv_part, void* stream);
/* the same on the rows of the bf16 forward (HEPT_PREC_BF16: qhat (H, N, 64 B), kvhat (H, N, 128 B)), one bf16 MFMA per
 * product: the opt-in 16-bit training mode.  The per-table gradient rows are bf16 as well: dq_part16 (Tl, N, H, 32) and
 * dkv_part16 (Tl, N, H, 64) of bf16; hept_bwd_reduce16 sums them (in f32) like hept_bwd_reduce. */
int hept_block_attn_bwd_bf16(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos,
                             const float* gacc, int N, int H, int D, int Tl, int B, void* dq_part16,
                             void* dkv_part16, void* stream);
int hept_bwd_reduce16(const void* dq_part16, const void* dkv_part16, int Tl, int N, int H, int D, int C,
                      const float* coords, int raw_size, float* dq, float* dk, float* dv, float* dcs,
                      float* d_sqrt_w, void* stream);
/* coords (N, C) + d_sqrt_w (H, C) (both may be NULL): also d_sqrt_w[h,c] = sum_n dcs[n,h,c] * coords[n,c], the
 * gradient that flows on into w_rpe.weight.  Rows at and after raw_size (the src variant's zero-filled padding,
 * raw_size = N otherwise) get zero gradients.  With d_sqrt_w the head of dq_part is overwritten (it serves as
 * scratch for the per-workgroup sums once it has been read; the sums are added in a fixed order). */
int hept_bwd_reduce(const float* dq_part, const float* dkv_part, int Tl, int N, int H, int D, int C,
                    const float* coords, int raw_size, float* dq, float* dk, float* dv, float* dcs,
                    float* d_sqrt_w, void* stream);
/* backward of hept_rpe_scale (example/hept.py:22-23,25 under autograd): d_w_rpe (H*D, (C-1)*K) from d_sqrt_w (H, C) */
int hept_rpe_scale_bwd(const float* w_rpe, const float* d_sqrt_w, int H, int D, int C, int K, float* d_w_rpe,
                       void* stream);
/* backward of hept_combine_out on table-summed f32 rows acc (N, H, 32) (example/hept.py:79-80 under autograd):
 * given g_out (N, D) writes gacc (N, H, 32) = gradient of acc, d_weight (D, H*D) and d_bias (D, may be NULL).
 * Any H <= 16, D <= 27 (the reference takes any, example/hept.py:34-41); D == 24 with H <= 8 takes the tuned kernels. */
/* scratch: hept_combine_bwd_scratch_bytes_shape(N, H, D) bytes (per-workgroup partial sums of d_weight / d_bias, added
 * in a fixed order by a second kernel: the gradients are bit-identical from run to run).
 * hept_combine_bwd_scratch_bytes(N) is the same for H = 8, D = 24. */
size_t hept_combine_bwd_scratch_bytes_shape(int N, int H, int D);
size_t hept_combine_bwd_scratch_bytes(int N);
int hept_combine_bwd(const float* acc, const float* g_out, const float* out_weight, int N, int H, int D,
                     float* gacc, float* d_weight, float* d_bias, void* scratch, size_t scratch_bytes,
                     void* stream);

/* SURVEY.md §8 f-1 — replaces prepare_input (example/transformer.py:35-63: per-cloud argsorts of eta / phi,
 * quantile_partition example/hept_utils.py:6-14, bit_shift x2 :10-13, pad_and_unpad :16-32 and the gathers by
 * pad_seq :59-62) for one batch of clouds whose points are contiguous.
 *   coords (n_raw, C) f32; cloud_start, pad_start: device i32 (n_clouds + 1) exclusive prefix sums of the raw /
 *   block-padded cloud sizes; regions (T, 2, H) f32; max_cloud = largest raw cloud; n_pad = pad_start[n_clouds].
 * Outputs: pad_seq (n_pad) i64, unpad (n_pad) u8 mask, coords_pad (n_pad, C) f32, codes_pad (T, H, n_pad) i64.
 * Ties (equal coordinates / equal codes) are broken by ascending index; the reference's argsort leaves them
 * undefined.  Requires every packed code < 2^24. */
size_t hept_prepare_workspace_bytes(int n_raw, int n_clouds, int max_cloud, int T, int H);
/* The host round trip in front of hept_prepare_input (example/transformer.py:35-43 synchronises on the cloud sizes as
 * well): one small kernel reads the sorted batch vector (i64 or i32, n_raw entries, cloud ids 0 .. n_clouds - 1) and
 * regions (T, 2, H), leaves cloud_start / pad_start (257 i32 each, device) for hept_prepare_input and writes a record
 * of 8 i32 into pinned host memory: [n_clouds, n_pad, longest cloud, smallest cloud (< 1: a cloud id without points),
 * overflow (more than 255 clouds: not resolved, use another path), f32 bits of the largest region count of axis 0,
 * of axis 1, 0x600DF00D].  The caller synchronises the stream and reads the record.
 * HARD PRECONDITION: host_record is pinned, device-mapped host memory (hipHostMalloc / hipHostRegister); ordinary
 * pageable memory is refused with HEPT_ERR_ARG (a kernel store to it would be a GPU fault, not an error code). */
int hept_prepare_probe(const void* batch, int batch_is_i64, int n_raw, int B, const float* regions, int T, int H,
                       int32_t* cloud_start, int32_t* pad_start, int32_t* host_record, void* stream);
int hept_prepare_input(const float* coords, int C, const int32_t* cloud_start, const int32_t* pad_start,
                       int n_clouds, int n_raw, int max_cloud, int n_pad, const float* regions, int T, int H,
                       int B, void* workspace, size_t workspace_bytes, int64_t* pad_seq, unsigned char* unpad,
                       float* coords_pad, int64_t* codes_pad, void* stream);

/* SURVEY.md §8 f-3, caller side — replaces the preparation of the reference's src variant
 * (src/models/baselines/transformer.py:43-57: pad_to_multiple x2, argsort(eta), argsort(phi), quantile_partition x2
 * src/models/model_utils/hash_utils.py:14-22, zeroing of the padded coordinates) for its single cloud.
 *   x (raw_size, F) f32 or NULL; coords (raw_size, C) f32; regions (T, 2, H) f32; N = raw_size padded to a block multiple.
 * Outputs: x_pad (N, F) (zero rows after raw_size; skipped when x is NULL), coords_pad (N, C) (zero rows after
 * raw_size), eta_idx / phi_idx (T*H, N) f32 = kwargs["region_indices"] (rows ordered (table, head)); the padding
 * slots rank last in index order.  Ties between equal coordinates are broken by ascending index. */
size_t hept_prepare_src_workspace_bytes(int N);
int hept_prepare_input_src(const float* x, int F, const float* coords, int C, int raw_size, int N,
                           const float* regions, int T, int H, void* workspace, size_t workspace_bytes,
                           float* x_pad, float* coords_pad, float* eta_idx, float* phi_idx, void* stream);

/* The small dense pieces around the operator in the TRAINING step of the Attn block (example/transformer.py:154-165
 * under autograd; csrc/block_train.hip).  Activation rows are D = J = 24 floats.  Reductions over the points are
 * two-stage with a fixed association: results do not depend on scheduling.
 *   hept_rows_wgrad   d_weight (O, 24) = dY^T . X and d_bias (O) = column sums of dY (NULL: skipped): the weight
 *                     gradient of a Linear(24 -> O) with input rows X (N, 24) and output gradient dY (N, O)
 *   hept_ln_bwd       LayerNorm(24) backward from x and dxn = d LayerNorm(x): dx, the normalised rows xn, d_ln_w, d_ln_b
 *   hept_ln_ffn_fwd   out = ff.2(relu(ff.0(norm2(x1))))            (:162; the residual and the dropouts stay outside)
 *   hept_ln_ffn_bwd   its backward: d_x1 and the gradients of norm2.weight/bias, ff.0.weight/bias, ff.2.weight/bias */
size_t hept_rows_wgrad_scratch_bytes(int N, int O);
int hept_rows_wgrad(const float* dY, const float* X, int N, int O, int J, float* d_weight, float* d_bias,
                    void* scratch, size_t scratch_bytes, void* stream);
size_t hept_ln_scratch_bytes(int N);
int hept_ln_bwd(const float* x, const float* dxn, const float* ln_w, const float* ln_b, float eps, int N, int D,
                float* dx, float* xn, float* d_ln_w, float* d_ln_b, void* scratch, size_t scratch_bytes, void* stream);
int hept_ln_ffn_fwd(const float* x1, const float* ln_w, const float* ln_b, float eps, const float* w1, const float* b1,
                    const float* w2, const float* b2, int N, int D, float* out, void* stream);
size_t hept_ln_ffn_bwd_scratch_bytes(int N);
int hept_ln_ffn_bwd(const float* x1, const float* d_out, const float* ln_w, const float* ln_b, float eps,
                    const float* w1, const float* b1, const float* w2, const float* b2, int N, int D, float* d_x1,
                    float* d_ln_w, float* d_ln_b, float* d_w1, float* d_b1, float* d_w2, float* d_b2, void* scratch,
                    size_t scratch_bytes, void* stream);

/* Optional stage timing with HIP events recorded on the caller's stream inside hept_forward /
 * hept_forward_partial (nothing like it exists in the reference; used by bench.py for the roofline).
 * mode 0: off (default).  mode 1: bracket the block_attn kernel only (2 events per call).
 * mode 2: bracket every stage (5 events per hept_forward call, 7 per hept_forward_sharded call).  `max_calls`
 * sizes the event pool; calls beyond it are not recorded.  hept_profile_read waits for the recorded events, adds
 * up the elapsed milliseconds per stage over the recorded calls into ms[7] -- [prep (with the RPE weight math),
 * sort, block_attn, combine/reduce, 0, 0] for hept_forward, [prep, sort, block_attn (all head groups, with the pushes
 * they carry), exposed push or transfer of the last head group, combine (+ output slice to the ranks), output
 * gather] for hept_forward_sharded; ms[6] (mode 2) = the FIRST of the sort's two launches (chunk sort; 0 for clouds
 * short enough for the one-launch sort), so that ms[1] - ms[6] is the second (bucket sort + v-row riders) -- stores
 * the number of calls in *n_calls and resets the pool. */
int hept_profile_enable(int mode, int max_calls);
int hept_profile_read(float* ms, int* n_calls);
/* Bracket only every `stride`-th forward call (default 1): an event pair costs a few microseconds of
 * stream time, so a timed loop samples the kernel instead of instrumenting every step. */
int hept_profile_stride(int stride);

#ifdef __cplusplus
}
#endif
#endif /* HEPT_HIP_H */
