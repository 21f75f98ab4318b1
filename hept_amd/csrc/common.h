// Shared device/host helpers for the gfx950 HEPT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hept_hip.h"

#define HEPT_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Which heads a block_attn launch covers and where their partial rows go: heads [h0, h0 + hg) of the H input slabs;
// the row of (table t, point n, head h) is written at row index t * tstride_rows + n * hout + (h - hsub) of `part`.
// Whole operator: {0, H, H, 0, N * H}.  Table sharding sends head groups one at a time: hout = hg, hsub = h0 makes
// the launch write a (n_rows, hg, row) buffer of its own.
struct HeadRange {
    int h0, hg, hout, hsub;
    long long tstride_rows;
    // f32 rows (HEPT_PREC_F32), round 5: the value rows are read where the caller left them -- v (N, H * D) f32, the
    // 1.0 of column D supplied by the kernel, rows >= raw_size zero -- instead of from the v half of the kvhat rows,
    // which then nobody builds (null: the kvhat rows carry v, as in every other mode)
    const float* vsrc;
    int raw_size;
    // workgroup -> (table, block) map of a block_attn launch (blockIdx % hg is always the head): tl > 0: the tables of a
    // block run side by side (t fastest), so that the three gathers of a point's rows meet in the XCD's L2; 0: table-major
    int tl;
};
struct VSrc {
    const float* v = nullptr;
    int raw_size = 0;
};

static inline int hept_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HEPT_OK : HEPT_ERR_LAUNCH;
}

// Kernels that need more than 64 KiB of dynamic LDS have to be told so once per device (the attribute belongs to the
// device's copy of the function): every call site keeps one flag per device ordinal.
#define HEPT_MAX_DEVICES 64
struct LdsRaised {
    bool done[HEPT_MAX_DEVICES] = {};
};
static inline int hept_raise_lds(LdsRaised& flags, const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HEPT_MAX_DEVICES) return HEPT_ERR_LAUNCH;
    if (flags.done[dev]) return HEPT_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return HEPT_ERR_LAUNCH;
    flags.done[dev] = true;
    return HEPT_OK;
}

// Deterministic second stage of a two-stage reduction: out[o] = sum_g partial[g * pitch + o] with a FIXED association
// (slice s of 16 takes g = s, s + 16, ... in ascending order, the 16 slice sums are added in ascending order), so the
// result does not depend on how the first stage's workgroups were scheduled.  256 threads = 16 outputs x 16 slices.
#define HEPT_FSUM_OUT 16
#define HEPT_FSUM_SLICES 16
__device__ __forceinline__ float hept_fixed_sum(const float* __restrict__ partial, int n_parts, size_t pitch, int o,
                                               bool valid, float* red_s /* [SLICES][OUT] */) {
    const int ol = threadIdx.x % HEPT_FSUM_OUT, sl = threadIdx.x / HEPT_FSUM_OUT;
    float acc = 0.f;
    if (valid)
        for (int g = sl; g < n_parts; g += HEPT_FSUM_SLICES) acc += partial[(size_t)g * pitch + o];
    red_s[sl * HEPT_FSUM_OUT + ol] = acc;
    __syncthreads();
    float tot = 0.f;
    if (sl == 0)
        for (int s2 = 0; s2 < HEPT_FSUM_SLICES; ++s2) tot += red_s[s2 * HEPT_FSUM_OUT + ol];
    return tot;
}

// float -> bf16, round to nearest even, on the hardware converter (v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) float hept_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 hept_bf16x2;
__device__ __forceinline__ unsigned int hept_pack_bf16(float lo, float hi) {
    const hept_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hept_bf16x2));
}
// the two bf16 halves of a packed dword, widened back to fp32
__device__ __forceinline__ float hept_bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hept_bf16_hi(unsigned int w) { return __uint_as_float(w & 0xFFFF0000u); }

// float pair -> packed fp16 (round to nearest even, v_cvt_pk_f16_f32), saturating at the largest finite fp16
typedef __attribute__((ext_vector_type(2))) _Float16 hept_f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned int hept_pack_f16(float lo, float hi) {
    const hept_f32x2 v = {fminf(fmaxf(lo, -65504.f), 65504.f), fminf(fmaxf(hi, -65504.f), 65504.f)};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hept_f16x2));
}
__device__ __forceinline__ float hept_f16_lo(unsigned int w) { return (float)__builtin_bit_cast(hept_f16x2, w)[0]; }
__device__ __forceinline__ float hept_f16_hi(unsigned int w) { return (float)__builtin_bit_cast(hept_f16x2, w)[1]; }

// Row of the 32x32 MFMA accumulator held in register r by lane-half hh
// (C/D layout of v_mfma_f32_32x32x*: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
__device__ __forceinline__ int hept_acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---- f32 products on the bf16 matrix pipe (block_attn.hip, block_attn_bwd.hip) ------------------------------------
// a = ah + am + al with ah = bf16(a), am = bf16(a - ah), al = bf16(a - ah - am); every residual is exact in f32.
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
__device__ __forceinline__ void split3_bf16(const float (&a)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int hp = hept_pack_bf16(a[2 * j], a[2 * j + 1]);
        const float r0 = a[2 * j] - hept_bf16_lo(hp), r1 = a[2 * j + 1] - hept_bf16_hi(hp);
        const unsigned int mp = hept_pack_bf16(r0, r1);
        const float s0 = r0 - hept_bf16_lo(mp), s1 = r1 - hept_bf16_hi(mp);
        h[j] = hp;
        m[j] = mp;
        l[j] = hept_pack_bf16(s0, s1);
    }
}
__device__ __forceinline__ void split2_bf16(const float (&a)[8], u32x4& h, u32x4& m) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int hp = hept_pack_bf16(a[2 * j], a[2 * j + 1]);
        h[j] = hp;
        m[j] = hept_pack_bf16(a[2 * j] - hept_bf16_lo(hp), a[2 * j + 1] - hept_bf16_hi(hp));
    }
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}

// pr = min(exp(x), 1) for the 16 logits of an accumulator (= exp(min(x, 0)): exp is monotone, exp(0) = 1).  The kernels
// issue a VALU instruction every other cycle they are resident (64 % VALU-busy at tracking-60k), and the plain form is
// three per logit (v_mul by log2(e), v_exp, v_min): here the multiply is packed (v_pk_mul_f32, two logits per
// instruction; the same IEEE product) and the upper bound is the clamp bit of v_exp itself ([0, 1]: exp is never
// negative) -- 1.5 instructions per logit, the same values.
__device__ __forceinline__ void exp_clamped(const f32x16& x, float (&pr)[16]) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x2_t t = f32x2_t{x[2 * j], x[2 * j + 1]} * f32x2_t{1.442695041f, 1.442695041f};   // 0x3FB8AA3B, as __expf
        pr[2 * j] = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(t[0]), 0.f), 1.f);
        pr[2 * j + 1] = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(t[1]), 0.f), 1.f);
    }
}

// ---- cache policy of the once-read / once-written streams (round 6) ------------------------------------------------------
// The forward's intermediates (q^ / k^|v rows, hashes, pairs, positions: ~140 MB with 16-bit rows) fit the 256 MB Infinity
// Cache, and the block attention gathers them at the cache's rate instead of HBM's -- as long as the streams that are
// touched exactly once (the caller's q, k, v: 138 MB; the sort's pairs on their way out; the partial rows) do not push
// them out.  Those streams are loaded / stored non-temporal (`nt`): tracking-60k bf16 block attention 82 -> 57 us with no
// other change (profiles/r06_experiments.txt).  Each switch is a build flag for A/B: -DHEPT_NT_<stream>=0.
#ifndef HEPT_NT_PREP_IN
#define HEPT_NT_PREP_IN 1     // row builder: q, k (and v) input tiles
#endif
#ifndef HEPT_NT_KA_IN
#define HEPT_NT_KA_IN 1       // chunk sort: the hashes
#endif
#ifndef HEPT_NT_KB_IN
#define HEPT_NT_KB_IN 1       // bucket sort: the bucket's pairs
#endif
#ifndef HEPT_NT_RIDER_IN32
#define HEPT_NT_RIDER_IN32 1  // v rows converted by the riders of the bucket-sort launch, f32 rows (block attention -6 us)
#endif
#ifndef HEPT_NT_RIDER_IN16
#define HEPT_NT_RIDER_IN16 0  // ... 16-bit rows: the riders' 16-B pieces of a line meet in L1, which nt loads bypass (sort +3 us)
#endif
#ifndef HEPT_NT_CODES
#define HEPT_NT_CODES 0       // chunk sort: the int64 AND codes (read twice: q and k segments)
#endif
#ifndef HEPT_NT_OUT
#define HEPT_NT_OUT 0         // combine: the (N, D) output rows
#endif
#ifndef HEPT_NT_BWD_ROWS
#define HEPT_NT_BWD_ROWS 0    // backward (f32 tiles): the per-table gradient rows (553 MB, read once by bwd_reduce)
#endif
template <bool NT, typename T>
__device__ __forceinline__ T hept_ld(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void hept_st(T* p, const T& v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- wave-wide scan / reductions on DPP (one VALU instruction per step; __shfl_up / __shfl_xor compile to ds_bpermute
// plus address arithmetic, ~4 VALU and an LDS round trip per step).  gfx9 DPP: row_shr within rows of 16 lanes, then
// row_bcast:15 (rows 1 and 3 take the last lane of the row before them) and row_bcast:31 (rows 2, 3 take lane 31).
__device__ __forceinline__ unsigned int hept_wave_scan_add(unsigned int x) {   // inclusive prefix sum over the 64 lanes
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1, 3
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ unsigned int hept_wave_sum(unsigned int x) {   // the same value in every lane
    return (unsigned int)__builtin_amdgcn_readlane((int)hept_wave_scan_add(x), 63);
}
__device__ __forceinline__ unsigned int hept_wave_max(unsigned int x) {
    auto mx = [](unsigned int a, unsigned int b) { return a > b ? a : b; };
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true));
    x = mx(x, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true));
    return (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
}

// ---- entry points shared between translation units (not part of the C ABI) ---------------------------------------------
// hept_prep_hash / hept_prep_hash_fused with the RPE weight math folded in (prep_hash.hip): K == 0: `sqrt_w` is
// sqrt_w (H, C); K > 0: it is w_rpe.weight (H*D, (C-1)*K) and the kernels compute the scale in their prologue.
int hept_prep_hash_rpe(const float* q, const float* k, const float* v, const float* coords, const float* sqrt_w, int K,
                       const float* alpha, const int64_t* codes, int N, int raw_size, int H, int D, int C, int T, int t0,
                       int Tl, int precision, void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax,
                       void* stream, int roles = 3,   // roles == 2: q and k rows + hashes only (the v rows: HeptRowsJob)
                       void* zero_ptr = nullptr, size_t zero_bytes = 0);   // scratch the launch clears on its way (hept_sort_zero_block)
int hept_prep_hash_fused_rpe(const float* x, const float* norm_w, const float* norm_b, float eps, const float* w_q,
                             const float* w_k, const float* w_v, const float* coords, const float* sqrt_w, int K,
                             const float* alpha, const int64_t* codes, int N, int raw_size, int H, int D, int C, int T,
                             int t0, int Tl, int precision, void* qhat, void* kvhat, float* qproj, float* kproj,
                             float* minmax, void* stream, void* zero_ptr = nullptr, size_t zero_bytes = 0);
// The v half of the kvhat rows written by workgroups that ride in the bucket-sort launch (sort_tables.hip) instead of by
// the row builder's third role: hept_prep_hash_rpe with roles == 2 leaves them out, hept_sort_tables_rows /
// hept_sort_tables_src_rows with a job description write them.  hept_sort_carries_rows: the sort of N-key segments has
// such a launch (segments longer than the one-workgroup sort) and the rows split into 16-B pieces (D % 4 == 0).
struct HeptRowsJob {
    const float* v;   // (N, H * D) fp32
    void* kvhat;
    int N, raw_size, H, D, precision;
};
bool hept_sort_carries_rows(int N, int H, int D);
// zeroed: the block hept_sort_zero_block names has been zeroed on this stream by the caller (the row builder of the same
// forward does it: hept_prep_hash_rpe's `zero` argument); otherwise the sort pays a fill launch for it
int hept_sort_tables_rows(const float* qproj, const float* kproj, const int64_t* codes, const float* minmax, int N, int H,
                          int T, int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos, const HeptRowsJob* rows,
                          void* stream, bool zeroed = false);
int hept_sort_tables_src_rows(const float* qproj, const float* kproj, const float* eta_idx, const float* phi_idx,
                              const float* cfac, float* minmax, int N, int H, int T, int t0, int Tl, void* sort_ws,
                              int32_t* qpos, int32_t* kpos, const HeptRowsJob* rows, void* stream, bool zeroed = false);
void hept_sort_zero_block(void* sort_ws, int N, int H, int Tl, void** ptr, size_t* bytes);
// stage timing (capi.hip, hept_profile_*): the sort calls this between its two launches
void hept_prof_mark_sort_mid(void* stream);
// hept_segmented_argsort for a caller that knows finite bounds of its keys (prepare.hip: packed code keys): the
// per-segment range pass (a fill and a kernel) is skipped; any bounds give the exact stable sort
int hept_segmented_argsort_bounded(const float* keys, int S, int L, float lo, float hi, void* ws, int32_t* pos,
                                   void* stream);
