// prep_hash: coordinate augmentation + E2LSH projection + hash range partials.
//
// Replaces, for one call (reference file:line):
//   prep_qk                    example/hept.py:21-28
//   "n h d -> h n d" views     example/hept.py:57-59
//   E2LSH.forward (bmm)        example/hept_utils.py:45-47
//   min / max of lsh_mapping   example/hept_utils.py:66-69
//
// HBM-bound streaming kernel: reads q,k,v (N,H*D) once, writes the
// head-major augmented rows that block_attn gathers later:
//   qhat  (H,N,32)  : [ q | sqrt_w*coords | 0 .. | -0.5*|q^|^2 (f32 bits in the last 4 bytes) ]
//   kvhat (H,N,64)  : [ k^ row as above | v | 1.0 at column D | 0 .. ]
// The 1.0 column makes the P.V MFMA also produce the row sum (denominator).
// For bf16 tiles the norm is taken of the *rounded* values, so that
// q.k - 0.5|q|^2 - 0.5|k|^2 is exactly -0.5|q_r - k_r|^2 <= 0 up to fp32 rounding.
// Hashes are always computed from the unrounded fp32 values.
#include "common.h"

namespace {

constexpr int PREP_THREADS = 256;
// 16-B slots of the wave-private LDS buffer: 64 input rows at pitch 7 (D = 24) or 64 output rows of up to
// 8 chunks + one pad slot per head
constexpr int PREP_WAVE_SLOTS = 64 * 8 + 8;
constexpr int PREP_POINTS = 8;   // points per wave iteration; lane = (point = lane >> 3, head = lane & 7)
#ifndef HEPT_PREP_CODE_SAMPLE
#define HEPT_PREP_CODE_SAMPLE 8   // the q role looks at the AND codes of one tile in this many (see prep_role)
#endif
// LDS pitch (floats) of one head's alpha slab [e][TMAX] (TMAX = 4 or 8 table slots, template parameter): = 4 (mod 32), so that the 8 heads of a wave read
// 8 disjoint bank groups (a plain E * 8 pitch put them on two groups: 4-way conflicts, 60 % of the LDS cycles)
constexpr int alpha_pitch(int E, int TMAX) { return ((E * TMAX + 27) / 32) * 32 + 4; }

// sqrt_w[h][c] = sqrt(2 * sum_k exp(min(sum_d w[h*D+d][r*K+k], 50))), column 0 duplicated (eta, phi share dR).
// One thread per (h, r, k) term (coalesced over k), then a K-term sum per (h, r).
__global__ __launch_bounds__(1024) void rpe_scale_kernel(const float* __restrict__ w, int H, int D, int C, int K,
                                                         float* __restrict__ sqrt_w) {
    __shared__ float term_s[1024];
    const int R = C - 1, RK = R * K, total = H * RK;
    const int i = threadIdx.x;
    if (i < total) {
        const int h = i / RK, rk = i - h * RK;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) s += w[(size_t)(h * D + d) * RK + rk];  // unrolled: loads in flight together
        term_s[i] = expf(fminf(s, 50.f));
    }
    __syncthreads();
    if (i < H * R) {
        const int h = i / R, r = i - h * R;
        float tot = 0.f;
        for (int kk = 0; kk < K; ++kk) tot += term_s[h * RK + r * K + kk];
        const float val = sqrtf(2.f * tot);
        sqrt_w[h * C + r + 1] = val;
        if (r == 0) sqrt_w[h * C] = val;
    }
}

// The same weight math inside another kernel's prologue (the row builders below): every workgroup computes sqrt_w (H, C)
// of w_rpe.weight into its own LDS -- H*(C-1)*K <= 1024 terms of D L2-resident loads each, ~1 us beside the alpha
// staging -- so the operator never depends on a cached copy of a parameter the caller may update in place
// (reference example/hept.py:22-25 recomputes it every forward).  Same operations in the same order as
// rpe_scale_kernel: the two are bit-identical.  Ends with the values visible to the whole workgroup.
template <int NT>
__device__ __forceinline__ void rpe_scale_lds(const float* __restrict__ w, int H, int D, int C, int K,
                                              float* __restrict__ sw_s, float* __restrict__ term_s) {
    const int R = C - 1, RK = R * K, total = H * RK;
    // the loads of BOTH terms a thread may own (total <= 1024 = 4 NT at the least) are issued before the first sum: the
    // prologue is a chain of L2 round trips, and 8 loads in flight per thread made it five of them (~3 us per launch)
    constexpr int TPT = 1024 / NT;   // terms per thread at the largest supported total
    for (int i0 = threadIdx.x; i0 < total; i0 += NT * TPT) {
        float v[TPT][28];
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int i = i0 + u * NT;
            const int h = i / RK, rk = i - h * RK;
#pragma unroll
            for (int d = 0; d < 28; ++d)
                v[u][d] = (i < total && d < D) ? w[(size_t)(h * D + d) * RK + rk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int i = i0 + u * NT;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 28; ++d)
                if (d < D) s += v[u][d];      // ascending d, as rpe_scale_kernel
            if (i < total) term_s[i] = expf(fminf(s, 50.f));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * R; i += NT) {
        const int h = i / R, r = i - h * R;
        float tot = 0.f;
        for (int kk = 0; kk < K; ++kk) tot += term_s[h * RK + r * K + kk];
        const float val = sqrtf(2.f * tot);
        sw_s[h * C + r + 1] = val;
        if (r == 0) sw_s[h * C] = val;
    }
    __syncthreads();
}

// backward of rpe_scale (training): d w[h*D+d][r*K+k] = [a <= 50] exp(a) * sum_{c -> r} d sqrt_w[h][c] / sqrt_w[h][c]
// with a = sum_d w[h*D+d][r*K+k] (the same for every d of a head; torch's clamp passes the gradient at equality) and
// columns c = 0 and c = 1 both fed by r = 0.  sqrt_w = 0 (all terms underflowed) gives inf/nan exactly as autograd does.
__global__ __launch_bounds__(1024) void rpe_scale_bwd_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ d_sqrt_w, int H, int D, int C,
                                                             int K, float* __restrict__ d_w) {
    const int R = C - 1, RK = R * K, total = H * RK;
    const int i = threadIdx.x;
    __shared__ float term_s[1024];
    float a = 0.f;
    if (i < total) {
        const int h = i / RK, rk = i - h * RK;
#pragma unroll 8
        for (int d = 0; d < D; ++d) a += w[(size_t)(h * D + d) * RK + rk];
        term_s[i] = expf(fminf(a, 50.f));
    }
    __syncthreads();
    if (i < total) {
        const int h = i / RK, rk = i - h * RK, r = rk / K;
        float tot = 0.f;
        for (int kk = 0; kk < K; ++kk) tot += term_s[h * RK + r * K + kk];
        const float s = sqrtf(2.f * tot);  // = sqrt_w[h][r + 1]
        float g = d_sqrt_w[h * C + r + 1];
        if (r == 0) g += d_sqrt_w[h * C];
        const float da = a <= 50.f ? (g / s) * term_s[i] : 0.f;
        for (int d = 0; d < D; ++d) d_w[(size_t)(h * D + d) * RK + rk] = da;
    }
}

// A block of scratch the NEXT kernel of the forward needs zeroed (the bucket counters of the sort, sort_tables.hip:
// RegionArgs): the launch's workgroups clear it on their way in, a few stores each, instead of a fill launch.
struct ZeroJob {
    unsigned int* ptr;
    unsigned int words;
};
__device__ __forceinline__ void zero_block(const ZeroJob& z) {
    const unsigned int n_threads = gridDim.x * gridDim.y * blockDim.x;
    for (unsigned int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < z.words; i += n_threads)
        z.ptr[i] = 0u;
}

// Streaming transform, one lane per (point, head) row, three wave ROLES selected by blockIdx.y:
//   role 0 (q): q row + coords -> q^ row, q hashes, hash min/max, largest AND code
//   role 1 (k): k row + coords -> k^ half of the kvhat row, k hashes, hash min/max
//   role 2 (v): v row          -> v half of the kvhat row ([v | 1.0 at column D | 0]); not launched (roles == 2) when the
//               bucket-sort launch of the same forward writes these rows (sort_tables.hip: RowsJob)
// A wave covers 8 consecutive points x 8 heads = one contiguous 6-KiB run of the (N, H*D) input and, per
// head, one contiguous run of 8 output rows.  Both sides go through a wave-private LDS buffer so that
// every global access is a full-width, fully coalesced 16 B per lane (LDS executes one wave's accesses
// in order: no barrier): chunks -> padded rows (odd 16-B pitch: conflict-free) -> lane m = p*8 + h owns
// row (p, h); finished rows -> [head][point] order (+1 slot per head: conflict-free) -> linear chunks.
// Splitting the roles keeps every wave light (few registers, many waves per CU).
// FUSED: x is the (N, D) input of the Attn block instead of a (N, H*D) projection: every lane
// applies LayerNorm to its point's row (example/transformer.py:155) and computes ITS head's D outputs of the
// bias-free projection (w_q / w_k / w_v, :156) from the LDS-resident weight slab of this role -- q, k, v never
// exist in HBM.
struct FusedIn {
    const float* ln_w;   // norm1.weight (D)
    const float* ln_b;   // norm1.bias (D)
    const float* w_s;    // LDS: this role's projection weight, [h][j][d] (transposed) at head pitch FUSED_WPITCH
    float eps;
};
constexpr int FUSED_WPITCH = 24 * 24 + 4;  // head pitch = 4 (mod 32) dwords: the 8 heads' 16-B reads hit 8 distinct bank groups

template <int D, int C, int TILE, int ROLE, int TMAX, bool FUSED = false>
__device__ __forceinline__ void prep_role(const float* __restrict__ x, const float* __restrict__ coords,
                                          const float* __restrict__ sw_s, const float* __restrict__ alpha_s,
                                          const int64_t* __restrict__ codes, int N, int raw_size, int t0, int Tl,
                                          char* __restrict__ out_rows, float* __restrict__ proj,
                                          float* __restrict__ red_s, f32x4* __restrict__ tile_s,
                                          float* __restrict__ minmax, int slot, unsigned int* __restrict__ cmax_s,
                                          FusedIn fin = FusedIn{}) {
    constexpr int H = 8, E = D + C, HD = H * D, D4 = D / 4;
    constexpr int WAVES = PREP_THREADS / HEPT_WAVE;
    constexpr bool BF16 = TILE != HEPT_PREC_F32;      // 16-bit tiles
    constexpr bool F16QK = TILE == HEPT_PREC_MIXED16;  // q^/k^ rows in fp16 instead of bf16
    constexpr int QROW = BF16 ? 64 : 128;
    constexpr int ROWB = ROLE == 0 ? QROW : 2 * QROW;   // row pitch of the destination array
    constexpr int ROWOFF = ROLE == 2 ? QROW : 0;        // v lives in the second half of a kvhat row
    constexpr int CH = QROW / 16;                       // 16-B chunks per finished row
    constexpr int LOADS = PREP_POINTS * HD / 4 / HEPT_WAVE;  // input chunks per lane (6)
    constexpr int ROW4 = D4 | 1;                        // LDS pitch of an input row in 16-B slots (odd)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = lane & 7, p = lane >> 3;
    // FUSED rows come from registers: only the output image (64 rows of CH chunks + one pad slot per head) is staged
    constexpr int WSLOTS = FUSED ? 64 * CH + 8 : PREP_WAVE_SLOTS;
    f32x4* buf = tile_s + (size_t)w * WSLOTS;
    int wslot[LOADS];  // chunk c of the tile belongs to row c / D4 (= the lane that reads it), slot c % D4
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int c = j * 64 + lane;
        wslot[j] = (c / D4) * ROW4 + (c % D4);
    }

    // ... and, where the LDS has room (not FUSED: there the per-wave partials replace red_s), the hash range as well: a wave-
    // private [min | max] word per (table, head), LDS float atomics per tile instead of 2 TMAX loop-carried registers
    constexpr bool RANGE_LDS = !FUSED;
    float mn[RANGE_LDS ? 1 : TMAX], mx[RANGE_LDS ? 1 : TMAX];
    float* rgw = red_s + w * 2 * TMAX * H;   // RANGE_LDS: red_s is [wave][min | max][TMAX][H]
    if constexpr (RANGE_LDS) {
        for (int i = lane; i < 2 * TMAX * H; i += 64) rgw[i] = i < TMAX * H ? INFINITY : -INFINITY;
    } else {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) { mn[t] = INFINITY; mx[t] = -INFINITY; }
    }
    // the largest code seen lives in LDS, not in TMAX loop-carried registers (the tile loop sits on the 128-VGPR boundary):
    // a wave-private word per (table, head), raised by an LDS atomic in the tiles that look at the codes (codes are
    // >= 0: their float bit patterns order like the values).  A wave's LDS operations execute in order: no barrier.
    unsigned int* cmw = cmax_s + w * TMAX * H;
    if (ROLE == 0) {
        for (int i = lane; i < TMAX * H; i += 64) cmw[i] = 0u;
    }

    const int ntiles = (N + PREP_POINTS - 1) / PREP_POINTS;
    for (int tile_i = blockIdx.x * WAVES + w; tile_i < ntiles; tile_i += gridDim.x * WAVES) {
        const int n0 = tile_i * PREP_POINTS;
        const int rows = min(PREP_POINTS, N - n0);
        const int n = n0 + p;
        const bool live = p < rows;
        f32x4 xr[D4];
        if constexpr (FUSED) {
            float xv[D];
            const f32x4* xs = reinterpret_cast<const f32x4*>(x + (size_t)(live ? n : n0) * D);
#pragma unroll
            for (int j = 0; j < D4; ++j) {
                const f32x4 v4 = xs[j];
                xv[4 * j] = v4[0]; xv[4 * j + 1] = v4[1]; xv[4 * j + 2] = v4[2]; xv[4 * j + 3] = v4[3];
            }
            float mean = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) mean += xv[j];
            mean *= 1.0f / D;
            float var = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) { const float dlt = xv[j] - mean; var = fmaf(dlt, dlt, var); }
            const float rstd = 1.0f / sqrtf(var * (1.0f / D) + fin.eps);
#pragma unroll
            for (int j = 0; j < D; ++j) xv[j] = (xv[j] - mean) * rstd * fin.ln_w[j] + fin.ln_b[j];
            // out[d] = sum_j W[d][j] xn[j] with the slab stored [j][d]: one 16-B read brings W[d..d+3][j], and the four
            // outputs advance with two packed fp32 FMAs (v_pk_fma_f32) instead of four scalar ones
            const float* wrow = fin.w_s + h * FUSED_WPITCH;
#pragma unroll
            for (int g = 0; g < D4; ++g) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + j * D + 4 * g);
                    acc = __builtin_elementwise_fma(wv, f32x4{xv[j], xv[j], xv[j], xv[j]}, acc);
                }
                xr[g] = acc;
                // keep the 144 weight reads from being hoisted in front of the fma chains (they would need ~600 VGPRs)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)n0 * HD);
            const int valid_chunks = rows * (HD / 4);
            f32x4 xin[LOADS];
#pragma unroll
            for (int j = 0; j < LOADS; ++j)
                xin[j] = (j * 64 + lane < valid_chunks) ? hept_ld<HEPT_NT_PREP_IN>(src + j * 64 + lane) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < LOADS; ++j) buf[wslot[j]] = xin[j];
#pragma unroll
            for (int j = 0; j < D4; ++j) xr[j] = buf[lane * ROW4 + j];
        }
        // src variant (src/models/attention/hept.py:89-96): rows >= raw_size are padding: q^ = k^ = v = 0,
        // they take part in the hash range with projection 0 and then hash to +inf (sorted last)
        const bool is_pad = n >= raw_size;
        if (is_pad) {
#pragma unroll
            for (int j = 0; j < D4; ++j) xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float a[32];
        if (ROLE != 2) {
            float cs[C];
#pragma unroll
            for (int c = 0; c < C; ++c) cs[c] = (live && !is_pad) ? coords[(size_t)n * C + c] : 0.f;
            if (ROLE == 0 && codes && (__builtin_amdgcn_readfirstlane(tile_i) % HEPT_PREP_CODE_SAMPLE) == 0) {   // (wave-uniform)
                // largest AND code of this (table, head), from every HEPT_PREP_CODE_SAMPLE-th tile: it only sets the
                // scale of the sort's bucket ids (keys beyond the bound share the last id: the sort is exact for ANY
                // value here), so a sample's maximum serves as well as the true one -- and the row builder reads 1.4 MB
                // of int64 codes instead of 11.5 MB at tracking-60k.  Deterministic (tiles, not random numbers).
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
                    if (t < Tl && live)
                        atomicMax(&cmw[t * H + h], __float_as_uint(fmaxf(__ll2float_ru(codes[((size_t)(t0 + t) * H + h) * N + n]), 0.f)));
            }
#pragma unroll
            for (int c = 0; c < C; ++c) a[D + c] = sw_s[h * C + c] * cs[c];
#pragma unroll
            for (int e = E; e < 32; ++e) a[e] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < D4; ++j) { a[4 * j] = xr[j][0]; a[4 * j + 1] = xr[j][1]; a[4 * j + 2] = xr[j][2]; a[4 * j + 3] = xr[j][3]; }
        // finished row -> LDS slot of (head, point); the copy-out below turns slots into linear chunks
        char* dst = reinterpret_cast<char*>(buf + (h * PREP_POINTS + p) * CH + h);

        if (ROLE == 2) {
            a[D] = 1.f;
#pragma unroll
            for (int d = D + 1; d < 32; ++d) a[d] = 0.f;
            if (BF16) {
                u32x4* d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d4[j] = u32x4{hept_pack_bf16(a[8 * j], a[8 * j + 1]), hept_pack_bf16(a[8 * j + 2], a[8 * j + 3]),
                                  hept_pack_bf16(a[8 * j + 4], a[8 * j + 5]), hept_pack_bf16(a[8 * j + 6], a[8 * j + 7])};
            } else {
                f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
                for (int j = 0; j < 8; ++j) d4[j] = f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
            }
        } else {

        // E2LSH projections from the unrounded fp32 row: one ascending-e fma chain per table.  The slab is [e][TMAX]:
        // one 16-B LDS read brings alpha[e] of four tables, and the TMAX chains advance side by side (slots t >= Tl
        // hold zeros and are not stored)
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        // (the slab does not change between tiles: hide that from the compiler, or it keeps all 4 E words of it in
        //  registers across the tile loop)
        int al_off = h * alpha_pitch(E, TMAX);
        asm volatile("" : "+v"(al_off));
        const float* al_row = alpha_s + al_off;
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int g = 0; g < TMAX / 4; ++g) {
                const f32x4 al = *reinterpret_cast<const f32x4*>(al_row + e * TMAX + 4 * g);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[4 * g + u] = fmaf(a[e], al[u], acc[4 * g + u]);
            }
            // a few reads in flight are enough: pin the chains here, or all E reads are issued first and their
            // 4 E destination registers stay live until the fmas that were sunk below them
            if (e % (TMAX == 4 ? 6 : 3) == (TMAX == 4 ? 5 : 2)) {
#pragma unroll
                for (int t = 0; t < TMAX; ++t) asm volatile("" : "+v"(acc[t]));
            }
        }
        // (the range update is unconditional in t -- slots t >= Tl are never read -- so that the chains above stay one
        //  straight-line block instead of being sunk into TMAX branches with every alpha word live)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if constexpr (RANGE_LDS) {
                if (t < Tl && live) {
                    __hip_atomic_fetch_min(&rgw[t * H + h], acc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_max(&rgw[TMAX * H + t * H + h], acc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            } else {
                mn[t] = fminf(mn[t], live ? acc[t] : INFINITY);
                mx[t] = fmaxf(mx[t], live ? acc[t] : -INFINITY);
            }
            if (t < Tl && live) proj[((size_t)t * H + h) * N + n] = is_pad ? INFINITY : acc[t];
        }
        float ss = 0.f;
        if (BF16) {
            unsigned int wd[16];
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                wd[i] = F16QK ? hept_pack_f16(a[2 * i], a[2 * i + 1]) : hept_pack_bf16(a[2 * i], a[2 * i + 1]);
                // norm of the ROUNDED values
                const float r0 = F16QK ? hept_f16_lo(wd[i]) : hept_bf16_lo(wd[i]);
                const float r1 = F16QK ? hept_f16_hi(wd[i]) : hept_bf16_hi(wd[i]);
                ss = fmaf(r0, r0, ss);
                ss = fmaf(r1, r1, ss);
            }
            wd[15] = __float_as_uint(-0.5f * ss);
            u32x4* d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll
            for (int j = 0; j < 4; ++j) d4[j] = u32x4{wd[4 * j], wd[4 * j + 1], wd[4 * j + 2], wd[4 * j + 3]};
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) ss = fmaf(a[e], a[e], ss);
            a[31] = -0.5f * ss;
            f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
            for (int j = 0; j < 8; ++j) d4[j] = f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
        }
        }  // ROLE != 2

        // copy-out: chunk c of the [head][point][CH] image -> 8 consecutive destination rows per head
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = j * 64 + lane;
            const int hd = c / (PREP_POINTS * CH), rest = c % (PREP_POINTS * CH);
            const int pt = rest / CH, u = rest % CH;
            const f32x4 val = buf[c + hd];
            if (pt < rows)
                *reinterpret_cast<f32x4*>(out_rows + ((size_t)hd * N + n0 + pt) * ROWB + ROWOFF + u * 16) = val;
        }
    }
    if (ROLE == 2) return;

    // per-workgroup partial hash range: fold the waves' partials (RANGE_LDS: they are in LDS already; else lane keeps one
    // head (h = lane & 7): fold the 8 point-lanes first.  FUSED: red_s lives in the wave buffers -- LDS is what limits that
    // kernel's occupancy -- so every wave has to be done with its tiles first)
    if constexpr (FUSED) __syncthreads();
    if constexpr (!RANGE_LDS) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < Tl) {
                float lo = mn[t], hi = mx[t];
#pragma unroll
                for (int off = 8; off <= 32; off <<= 1) {
                    lo = fminf(lo, __shfl_xor(lo, off));
                    hi = fmaxf(hi, __shfl_xor(hi, off));
                }
                if (p == 0) {
                    float* r = red_s + w * 2 * TMAX * H;
                    r[t * H + h] = lo;
                    r[TMAX * H + t * H + h] = hi;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < Tl * H; i += PREP_THREADS) {
        const int t = i / H, hh = i % H;
        float lo = INFINITY, hi = -INFINITY, c = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            const float* r = red_s + ww * 2 * TMAX * H;
            lo = fminf(lo, r[t * H + hh]);
            hi = fmaxf(hi, r[TMAX * H + t * H + hh]);
            if (ROLE == 0) c = fmaxf(c, __uint_as_float(cmax_s[(ww * TMAX + t) * H + hh]));
        }
        // layout [Tl][H][HEPT_PREP_GRID][4]: the sort kernels reduce one (t,h) row with contiguous 16-B loads.  The
        // launch may have fewer workgroups per role than the role has slots: the spare slots get neutral values
        f32x4* mrow = reinterpret_cast<f32x4*>(minmax + ((size_t)t * H + hh) * HEPT_PREP_GRID * 4);
        mrow[slot] = f32x4{lo, hi, c, 0.f};
        for (int s2 = slot + (int)gridDim.x; s2 < (slot / (HEPT_PREP_GRID / 2) + 1) * (HEPT_PREP_GRID / 2); s2 += (int)gridDim.x)
            mrow[s2] = f32x4{INFINITY, -INFINITY, 0.f, 0.f};
    }
}

constexpr int PREP_SLOTS_PER_ROLE = HEPT_PREP_GRID / 2;  // q and k roles fill the HEPT_PREP_GRID partial slots
// Workgroups per role of a launch (<= the slots of a role; spare slots get neutral values): 3 x 341 = 1023 workgroups are
// ONE round at four per CU -- 3 x 512 was a round and a half, and every q- / k-role workgroup pays the prologue (alpha
// slab, the RPE weight math): 57.4 -> 54.8 us at tracking-60k.  Short clouds launch only the workgroups that get a tile.
#ifndef HEPT_PREP_WGS
#define HEPT_PREP_WGS 341
#endif
// ... and 2 x 512 when the v rows are written elsewhere (roles == 2: riders of the bucket-sort launch, sort_tables.hip)
#ifndef HEPT_PREP_WGS2
#define HEPT_PREP_WGS2 512
#endif
constexpr int PREP_WGS_PER_ROLE = HEPT_PREP_WGS;
static_assert(PREP_WGS_PER_ROLE <= PREP_SLOTS_PER_ROLE && HEPT_PREP_WGS2 <= PREP_SLOTS_PER_ROLE, "every workgroup owns a partial slot");
inline int prep_wgs(int N, int roles = 3) {
    const int tiles = (N + PREP_POINTS - 1) / PREP_POINTS, per_wg = PREP_THREADS / HEPT_WAVE;
    const int want = (tiles + per_wg - 1) / per_wg, cap = roles == 2 ? HEPT_PREP_WGS2 : PREP_WGS_PER_ROLE;
    if (want <= cap) return want < 1 ? 1 : want;
    // The launch is one round of workgroups and ends with its slowest wave: with `cap` workgroups some waves take one
    // tile more than others (tracking-60k: 7504 tiles on 2048 waves = 1360 waves with 4 tiles, 688 with 3).  Take the
    // number of trips that cap implies and the FEWEST workgroups that still make it: every wave gets the same number of
    // tiles (469 workgroups x 4 waves x 4 tiles) and the round has fewer waves to share the CUs with.
    // (measured with two roles -- 2 x 469 instead of 2 x 512 workgroups: row builder 45.9-46.9 -> 44.0-45.3 us; three
    //  roles at 3 x 313 instead of 3 x 341 are a round of fewer, longer waves and lose ~1 us: they keep the cap)
    static const bool even = [] { const char* e = getenv("HEPT_PREP_UNEVEN"); return !(e && *e && *e != '0'); }();
    if (!even || roles != 2) return cap;
    const int trips = (tiles + cap * per_wg - 1) / (cap * per_wg);
    return (tiles + trips * per_wg - 1) / (trips * per_wg);
}

// (fp16 q^/k^ rows, 4 table slots: left alone the allocation takes 135 VGPRs = 3 waves per SIMD where the bf16 build
//  takes 128; held to 4 waves like it -- the LDS allows 4 workgroups per CU)
template <int D, int C, int TILE, int TMAX>
__global__ __launch_bounds__(PREP_THREADS)
__attribute__((amdgpu_waves_per_eu((TILE == HEPT_PREC_MIXED16 && TMAX == 4) ? 4 : 1, (TILE == HEPT_PREC_MIXED16 && TMAX == 4) ? 4 : 8)))
void prep_hash_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ coords, const float* __restrict__ sqrt_w, int K, const float* __restrict__ alpha,
    const int64_t* __restrict__ codes, int N, int raw_size, int T, int t0, int Tl, void* __restrict__ qhat_,
    void* __restrict__ kvhat_, float* __restrict__ qproj, float* __restrict__ kproj, float* __restrict__ minmax,
    ZeroJob zero) {
    constexpr int H = 8, E = D + C;
    static_assert(D % 4 == 0 && E <= 30 && D <= 28, "row packing needs D%4==0, E<=30");
    zero_block(zero);
    __shared__ __attribute__((aligned(16))) float alpha_s[H * alpha_pitch(E, TMAX)];
    __shared__ float sw_s[H * C];
    __shared__ float red_s[(PREP_THREADS / HEPT_WAVE) * 2 * TMAX * H];   // per wave [min | max][TMAX][H]
    __shared__ f32x4 tile_s[(PREP_THREADS / HEPT_WAVE) * PREP_WAVE_SLOTS];
    __shared__ unsigned int cmax_s[(PREP_THREADS / HEPT_WAVE) * TMAX * H];
    static_assert(64 * ((D / 4) | 1) <= PREP_WAVE_SLOTS, "input tile does not fit the wave buffer");
    const int role = blockIdx.y;
    if (role != 2) {
        for (int i = threadIdx.x; i < H * E * TMAX; i += PREP_THREADS) {
            const int t = i % TMAX, he = i / TMAX;
            alpha_s[(he / E) * alpha_pitch(E, TMAX) + (he % E) * TMAX + t] = (t < Tl) ? alpha[(size_t)he * T + t0 + t] : 0.f;
        }
        // K > 0: `sqrt_w` is w_rpe.weight and the scale is computed here (the wave buffers are free until the tile loop)
        if (K > 0) rpe_scale_lds<PREP_THREADS>(sqrt_w, H, D, C, K, sw_s, reinterpret_cast<float*>(tile_s));
        else
            for (int i = threadIdx.x; i < H * C; i += PREP_THREADS) sw_s[i] = sqrt_w[i];
        __syncthreads();
    }
    if (role == 0)
        prep_role<D, C, TILE, 0, TMAX>(q, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(qhat_), qproj,
                                 red_s, tile_s, minmax, blockIdx.x, cmax_s);
    else if (role == 1)
        prep_role<D, C, TILE, 1, TMAX>(k, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(kvhat_), kproj,
                                 red_s, tile_s, minmax, PREP_SLOTS_PER_ROLE + blockIdx.x, cmax_s);
    else
        prep_role<D, C, TILE, 2, TMAX>(v, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(kvhat_), nullptr,
                                 red_s, tile_s, minmax, 0, cmax_s);
}

// Attn-block front end: LayerNorm + the three projections fused into the row builder (D = 24 only)
// 16-bit rows, 4 table slots: 39.9 KB of LDS = 4 workgroups per CU, and the registers are held to that occupancy too
template <int C, int TILE, int TMAX>
__global__ __launch_bounds__(PREP_THREADS)
__attribute__((amdgpu_waves_per_eu((TILE != HEPT_PREC_F32 && TMAX == 4) ? (C == 4 ? 3 : 4) : 2, (TILE != HEPT_PREC_F32 && TMAX == 4) ? 4 : 3)))
void prep_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
    const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
    const float* __restrict__ coords, const float* __restrict__ sqrt_w, int K, const float* __restrict__ alpha,
    const int64_t* __restrict__ codes, int N, int raw_size, int T, int t0, int Tl, void* __restrict__ qhat_,
    void* __restrict__ kvhat_, float* __restrict__ qproj, float* __restrict__ kproj, float* __restrict__ minmax,
    ZeroJob zero) {
    constexpr int D = 24, H = 8, E = D + C;
    zero_block(zero);
    __shared__ __attribute__((aligned(16))) float alpha_s[H * alpha_pitch(E, TMAX)];
    __shared__ float sw_s[H * C];
    // wave buffers: the output image only (264 slots for 16-bit rows, 520 for f32 rows); red_s reuses them after the loop
    constexpr int WSLOTS = 64 * (TILE != HEPT_PREC_F32 ? 4 : 8) + 8;
    __shared__ f32x4 tile_s[(PREP_THREADS / HEPT_WAVE) * WSLOTS];
    float* red_s = reinterpret_cast<float*>(tile_s);
    static_assert((PREP_THREADS / HEPT_WAVE) * 2 * TMAX * H * 4 <= (PREP_THREADS / HEPT_WAVE) * WSLOTS * 16, "red_s fits the wave buffers");
    __shared__ __attribute__((aligned(16))) float w_s[H * FUSED_WPITCH];
    __shared__ unsigned int cmax_s[(PREP_THREADS / HEPT_WAVE) * TMAX * H];
    const int role = blockIdx.y;
    const float* wsrc = role == 0 ? wq : (role == 1 ? wk : wv);
    for (int i = threadIdx.x; i < H * D * D; i += PREP_THREADS) {  // W row (h, d), column j  ->  slab h, [j][d]
        const int hd = i / D, j = i % D;
        w_s[(hd / D) * FUSED_WPITCH + j * D + hd % D] = wsrc[i];
    }
    if (role != 2) {
        for (int i = threadIdx.x; i < H * E * TMAX; i += PREP_THREADS) {
            const int t = i % TMAX, he = i / TMAX;
            alpha_s[(he / E) * alpha_pitch(E, TMAX) + (he % E) * TMAX + t] = (t < Tl) ? alpha[(size_t)he * T + t0 + t] : 0.f;
        }
        if (K > 0) rpe_scale_lds<PREP_THREADS>(sqrt_w, H, D, C, K, sw_s, reinterpret_cast<float*>(tile_s));   // see prep_hash_kernel
        else
            for (int i = threadIdx.x; i < H * C; i += PREP_THREADS) sw_s[i] = sqrt_w[i];
    }
    __syncthreads();
    const FusedIn fin{ln_w, ln_b, w_s, eps};
    if (role == 0)
        prep_role<D, C, TILE, 0, TMAX, true>(x, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(qhat_),
                                       qproj, red_s, tile_s, minmax, blockIdx.x, cmax_s, fin);
    else if (role == 1)
        prep_role<D, C, TILE, 1, TMAX, true>(x, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(kvhat_),
                                       kproj, red_s, tile_s, minmax, PREP_SLOTS_PER_ROLE + blockIdx.x, cmax_s, fin);
    else
        prep_role<D, C, TILE, 2, TMAX, true>(x, coords, sw_s, alpha_s, codes, N, raw_size, t0, Tl, reinterpret_cast<char*>(kvhat_),
                                       nullptr, red_s, tile_s, minmax, 0, cmax_s, fin);
}

template <int C>
int launch_prep_fused(const float* x, const float* ln_w, const float* ln_b, float eps, const float* wq, const float* wk,
                      const float* wv, const float* coords, const float* sqrt_w, int K, const float* alpha,
                      const int64_t* codes, int N, int raw_size, int T, int t0, int Tl, int precision, void* qhat,
                      void* kvhat, float* qproj, float* kproj, float* minmax, hipStream_t st, ZeroJob zero) {
    const dim3 grid(prep_wgs(N), 3);
    // table slots of the kernel (accumulators, alpha slab): 4 for the usual 1-4 tables per call, else HEPT_MAX_TABLES
#define HEPT_FUSED_LAUNCH(TILE, TMAX)                                                                                  \
    hipLaunchKernelGGL((prep_fused_kernel<C, TILE, TMAX>), grid, dim3(PREP_THREADS), 0, st, x, ln_w, ln_b, eps, wq, wk, \
                       wv, coords, sqrt_w, K, alpha, codes, N, raw_size, T, t0, Tl, qhat, kvhat, qproj, kproj, minmax, zero)
#define HEPT_FUSED_TILE(TILE)                                                                                          \
    do {                                                                                                               \
        if (Tl <= 4) HEPT_FUSED_LAUNCH(TILE, 4);                                                                       \
        else HEPT_FUSED_LAUNCH(TILE, HEPT_MAX_TABLES);                                                                 \
    } while (0)
    if (precision == HEPT_PREC_BF16) HEPT_FUSED_TILE(HEPT_PREC_BF16);
    else if (precision == HEPT_PREC_MIXED16) HEPT_FUSED_TILE(HEPT_PREC_MIXED16);
    else HEPT_FUSED_TILE(HEPT_PREC_F32);
#undef HEPT_FUSED_TILE
#undef HEPT_FUSED_LAUNCH
    return hept_launch_status();
}

template <int D, int C>
int launch_prep(const float* q, const float* k, const float* v, const float* coords, const float* sqrt_w, int K,
                const float* alpha, const int64_t* codes, int N, int raw_size, int T, int t0, int Tl, int precision,
                void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax, hipStream_t st, int roles, ZeroJob zero) {
    // q- and k-role workgroups each write one of the HEPT_PREP_GRID partial slots the sort kernel reduces
    const dim3 grid(prep_wgs(N, roles), roles);
    // table slots of the kernel (accumulators, alpha slab): 4 for the usual 1-4 tables per call, else HEPT_MAX_TABLES
#define HEPT_PREP_LAUNCH(TILE, TMAX)                                                                                 \
    hipLaunchKernelGGL((prep_hash_kernel<D, C, TILE, TMAX>), grid, dim3(PREP_THREADS), 0, st, q, k, v, coords, sqrt_w, \
                       K, alpha, codes, N, raw_size, T, t0, Tl, qhat, kvhat, qproj, kproj, minmax, zero)
#define HEPT_PREP_TILE(TILE)                                                                                         \
    do {                                                                                                             \
        if (Tl <= 4) HEPT_PREP_LAUNCH(TILE, 4);                                                                      \
        else HEPT_PREP_LAUNCH(TILE, HEPT_MAX_TABLES);                                                                \
    } while (0)
    if (precision == HEPT_PREC_BF16) HEPT_PREP_TILE(HEPT_PREC_BF16);
    else if (precision == HEPT_PREC_MIXED16) HEPT_PREP_TILE(HEPT_PREC_MIXED16);
    else HEPT_PREP_TILE(HEPT_PREC_F32);
#undef HEPT_PREP_TILE
#undef HEPT_PREP_LAUNCH
    return hept_launch_status();
}

// ---- any head count / head dimension -------------------------------------------------------------------------------
// The reference takes any num_heads, h_dim and coords_dim (example/hept.py:34-41).  The kernels above are tuned for the
// shipped models' H = 8 and six (D, C) pairs; everything else (1 <= H <= 16, D <= 28, C >= 1, D + C <= 30 -- the rows
// are 32 columns wide) takes this kernel: the same arithmetic (ascending-e fmaf hash chain over the unrounded row,
// norm of the rounded row for 16-bit tiles), one thread per (point, head) with plain loads, the same three roles and
// the same outputs, including the per-workgroup hash-range partials the sort reduces.  Not tuned (its loads are
// strided by the row length); correctness first.
__device__ __forceinline__ unsigned int f32_ordered(float x) {
    const unsigned int u = __float_as_uint(x);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned int u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <int TILE>
__global__ __launch_bounds__(PREP_THREADS) void prep_generic_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ coords, const float* __restrict__ sqrt_w, int K, const float* __restrict__ alpha,
    const int64_t* __restrict__ codes, int N, int raw_size, int H, int D, int C, int T, int t0, int Tl,
    void* __restrict__ qhat_, void* __restrict__ kvhat_, float* __restrict__ qproj, float* __restrict__ kproj,
    float* __restrict__ minmax, ZeroJob zero) {
    constexpr bool BF16 = TILE != HEPT_PREC_F32;
    constexpr bool F16QK = TILE == HEPT_PREC_MIXED16;
    constexpr int QROW = BF16 ? 64 : 128;
    zero_block(zero);
    __shared__ float alpha_s[16 * 30 * HEPT_MAX_TABLES];              // [h][e][t]
    __shared__ unsigned int red_s[HEPT_MAX_TABLES * 16 * 3];          // ordered-uint min / max / code max per (t, h)
    __shared__ float sw_s[16 * 29];                                   // sqrt_w (H, C)
    __shared__ float term_s[1024];                                    // scratch of rpe_scale_lds
    const int role = blockIdx.y, tid = threadIdx.x, E = D + C, HD = H * D;
    const int ppw = PREP_THREADS / H;             // points per workgroup step; thread = (point slot, head), head fixed
    const int h = tid % H, slot = tid / H;
    const bool worker = slot < ppw;
    if (role != 2) {
        for (int i = tid; i < H * E * HEPT_MAX_TABLES; i += PREP_THREADS) {
            const int t = i % HEPT_MAX_TABLES, he = i / HEPT_MAX_TABLES;
            alpha_s[i] = t < Tl ? alpha[(size_t)he * T + t0 + t] : 0.f;
        }
        for (int i = tid; i < HEPT_MAX_TABLES * 16 * 3; i += PREP_THREADS)
            red_s[i] = (i % 3 == 0) ? f32_ordered(INFINITY) : (i % 3 == 1 ? f32_ordered(-INFINITY) : f32_ordered(0.f));
        if (K > 0) rpe_scale_lds<PREP_THREADS>(sqrt_w, H, D, C, K, sw_s, term_s);   // `sqrt_w` is w_rpe.weight (see prep_hash_kernel)
        else
            for (int i = tid; i < H * C; i += PREP_THREADS) sw_s[i] = sqrt_w[i];
        __syncthreads();
    }
    const float* x = role == 0 ? q : (role == 1 ? k : v);
    char* out_rows = reinterpret_cast<char*>(role == 0 ? qhat_ : kvhat_);
    const int rowb = role == 0 ? QROW : 2 * QROW, rowoff = role == 2 ? QROW : 0;
    float* proj = role == 0 ? qproj : kproj;
    float mn[HEPT_MAX_TABLES], mx[HEPT_MAX_TABLES], cm[HEPT_MAX_TABLES];
#pragma unroll
    for (int t = 0; t < HEPT_MAX_TABLES; ++t) { mn[t] = INFINITY; mx[t] = -INFINITY; cm[t] = 0.f; }
    for (int n = blockIdx.x * ppw + slot; worker && n < N; n += gridDim.x * ppw) {
        const bool is_pad = n >= raw_size;
        float a[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) a[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 28; ++j)
            if (j < D) a[j] = is_pad ? 0.f : x[(size_t)n * HD + h * D + j];
        char* dst = out_rows + ((size_t)h * N + n) * rowb + rowoff;
        if (role == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j == D) a[j] = 1.f;
        } else {
#pragma unroll
            for (int c = 0; c < 30; ++c)
                if (c < C) {
                    const float val = is_pad ? 0.f : sw_s[h * C + c] * coords[(size_t)n * C + c];
#pragma unroll
                    for (int e = 1; e < 30; ++e)
                        if (e == D + c) a[e] = val;
                }
            if (role == 0 && codes) {
#pragma unroll
                for (int t = 0; t < HEPT_MAX_TABLES; ++t)
                    if (t < Tl) cm[t] = fmaxf(cm[t], __ll2float_ru(codes[((size_t)(t0 + t) * H + h) * N + n]));
            }
#pragma unroll
            for (int t = 0; t < HEPT_MAX_TABLES; ++t) {
                if (t < Tl) {
                    float acc = 0.f;
#pragma unroll
                    for (int e = 0; e < 30; ++e)
                        if (e < E) acc = fmaf(a[e], alpha_s[(h * E + e) * HEPT_MAX_TABLES + t], acc);
                    proj[((size_t)t * H + h) * N + n] = is_pad ? INFINITY : acc;
                    mn[t] = fminf(mn[t], acc);
                    mx[t] = fmaxf(mx[t], acc);
                }
            }
        }
        if (BF16) {
            unsigned int wd[16];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool f16 = F16QK && role != 2;
                wd[i] = f16 ? hept_pack_f16(a[2 * i], a[2 * i + 1]) : hept_pack_bf16(a[2 * i], a[2 * i + 1]);
                const float r0 = f16 ? hept_f16_lo(wd[i]) : hept_bf16_lo(wd[i]);
                const float r1 = f16 ? hept_f16_hi(wd[i]) : hept_bf16_hi(wd[i]);
                if (i < 15) { ss = fmaf(r0, r0, ss); ss = fmaf(r1, r1, ss); }   // columns 30, 31 are zero (E <= 30)
            }
            if (role != 2) wd[15] = __float_as_uint(-0.5f * ss);
            u32x4* d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll
            for (int j = 0; j < 4; ++j) d4[j] = u32x4{wd[4 * j], wd[4 * j + 1], wd[4 * j + 2], wd[4 * j + 3]};
        } else {
            if (role != 2) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 30; ++e) ss = fmaf(a[e], a[e], ss);   // zeros beyond E add nothing
                a[31] = -0.5f * ss;
            }
            f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
            for (int j = 0; j < 8; ++j) d4[j] = f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
        }
    }
    if (role == 2) return;
    if (worker) {
#pragma unroll
        for (int t = 0; t < HEPT_MAX_TABLES; ++t)
            if (t < Tl) {
                atomicMin(&red_s[(t * 16 + h) * 3], f32_ordered(mn[t]));
                atomicMax(&red_s[(t * 16 + h) * 3 + 1], f32_ordered(mx[t]));
                atomicMax(&red_s[(t * 16 + h) * 3 + 2], f32_ordered(cm[t]));
            }
    }
    __syncthreads();
    const int pslot = (role == 0 ? 0 : HEPT_PREP_GRID / 2) + blockIdx.x;
    for (int i = tid; i < Tl * H; i += PREP_THREADS) {
        const int t = i / H, hh = i % H;
        const unsigned int* r = red_s + (t * 16 + hh) * 3;
        *reinterpret_cast<f32x4*>(minmax + (((size_t)t * H + hh) * HEPT_PREP_GRID + pslot) * 4) =
            f32x4{f32_from_ordered(r[0]), f32_from_ordered(r[1]), f32_from_ordered(r[2]), 0.f};
    }
}

int launch_prep_generic(const float* q, const float* k, const float* v, const float* coords, const float* sqrt_w, int K,
                        const float* alpha, const int64_t* codes, int N, int raw_size, int H, int D, int C, int T, int t0,
                        int Tl, int precision, void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax,
                        hipStream_t st, int roles, ZeroJob zero) {
    const dim3 grid(HEPT_PREP_GRID / 2, roles);   // q- and k-role workgroups each own one partial slot
    if (precision == HEPT_PREC_BF16)
        hipLaunchKernelGGL((prep_generic_kernel<HEPT_PREC_BF16>), grid, dim3(PREP_THREADS), 0, st, q, k, v, coords, sqrt_w,
                           K, alpha, codes, N, raw_size, H, D, C, T, t0, Tl, qhat, kvhat, qproj, kproj, minmax, zero);
    else if (precision == HEPT_PREC_MIXED16)
        hipLaunchKernelGGL((prep_generic_kernel<HEPT_PREC_MIXED16>), grid, dim3(PREP_THREADS), 0, st, q, k, v, coords, sqrt_w,
                           K, alpha, codes, N, raw_size, H, D, C, T, t0, Tl, qhat, kvhat, qproj, kproj, minmax, zero);
    else
        hipLaunchKernelGGL((prep_generic_kernel<HEPT_PREC_F32>), grid, dim3(PREP_THREADS), 0, st, q, k, v, coords, sqrt_w,
                           K, alpha, codes, N, raw_size, H, D, C, T, t0, Tl, qhat, kvhat, qproj, kproj, minmax, zero);
    return hept_launch_status();
}

}  // namespace

extern "C" int hept_rpe_scale(const float* w_rpe, int H, int D, int C, int K, float* sqrt_w, void* stream) {
    if (!w_rpe || !sqrt_w) return HEPT_ERR_ARG;
    if (H < 1 || D < 1 || C < 2 || K < 1 || H * (C - 1) * K > 1024) return HEPT_ERR_SHAPE;
    hipLaunchKernelGGL(rpe_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w_rpe, H, D, C, K, sqrt_w);
    return hept_launch_status();
}

extern "C" int hept_rpe_scale_bwd(const float* w_rpe, const float* d_sqrt_w, int H, int D, int C, int K, float* d_w_rpe,
                                  void* stream) {
    if (!w_rpe || !d_sqrt_w || !d_w_rpe) return HEPT_ERR_ARG;
    if (H < 1 || D < 1 || C < 2 || K < 1 || H * (C - 1) * K > 1024) return HEPT_ERR_SHAPE;
    hipLaunchKernelGGL(rpe_scale_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w_rpe, d_sqrt_w, H, D, C, K,
                       d_w_rpe);
    return hept_launch_status();
}

// internal (common.h): K == 0: `sqrt_w` is sqrt_w (H, C); K > 0: it is w_rpe.weight (H*D, (C-1)*K) and every workgroup
// of the launch computes the scale in its prologue (rpe_scale_lds)
int hept_prep_hash_rpe(const float* q, const float* k, const float* v, const float* coords, const float* sqrt_w, int K,
                       const float* alpha, const int64_t* codes, int N, int raw_size, int H, int D, int C, int T, int t0,
                       int Tl, int precision, void* qhat, void* kvhat, float* qproj, float* kproj, float* minmax,
                       void* stream, int roles, void* zero_ptr, size_t zero_bytes) {
    if (roles != 2 && roles != 3) return HEPT_ERR_ARG;
    if (zero_bytes % 4 != 0 || (zero_bytes && !zero_ptr) || zero_bytes > 0xFFFFFFFFull) return HEPT_ERR_ARG;
    const ZeroJob zero{reinterpret_cast<unsigned int*>(zero_ptr), (unsigned int)(zero_bytes / 4)};
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    if (K < 0 || (K > 0 && (C < 2 || H * (C - 1) * K > 1024))) return HEPT_ERR_SHAPE;
    if (!q || !k || !v || !coords || !sqrt_w || !alpha || !qhat || !kvhat || !qproj || !kproj || !minmax)
        return HEPT_ERR_ARG;
    if (H < 1 || H > 16 || D < 1 || D > 28 || C < 1 || D + C > 30) return HEPT_ERR_SHAPE;
    if (N < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (precision == HEPT_PREC_F32_MFMA || precision == HEPT_PREC_F32_DIFF) precision = HEPT_PREC_F32;  // same f32 tile rows, another block_attn kernel
    if (precision != HEPT_PREC_F32 && precision != HEPT_PREC_BF16 && precision != HEPT_PREC_MIXED16)
        return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define HEPT_PREP_CASE(DD, CC)                                                                             \
    if (H == 8 && D == DD && C == CC)                                                                      \
        return launch_prep<DD, CC>(q, k, v, coords, sqrt_w, K, alpha, codes, N, raw_size, T, t0, Tl, precision, \
                                   qhat, kvhat, qproj, kproj, minmax, st, roles, zero);
    HEPT_PREP_CASE(24, 6)
    HEPT_PREP_CASE(24, 4)
    HEPT_PREP_CASE(24, 2)
    HEPT_PREP_CASE(16, 6)
    HEPT_PREP_CASE(16, 4)
    HEPT_PREP_CASE(8, 4)
#undef HEPT_PREP_CASE
    return launch_prep_generic(q, k, v, coords, sqrt_w, K, alpha, codes, N, raw_size, H, D, C, T, t0, Tl, precision, qhat,
                               kvhat, qproj, kproj, minmax, st, roles, zero);
}

extern "C" int hept_prep_hash(const float* q, const float* k, const float* v, const float* coords,
                              const float* sqrt_w, const float* alpha, const int64_t* codes, int N, int raw_size,
                              int H, int D, int C, int T, int t0, int Tl, int precision, void* qhat, void* kvhat,
                              float* qproj, float* kproj, float* minmax, void* stream) {
    return hept_prep_hash_rpe(q, k, v, coords, sqrt_w, 0, alpha, codes, N, raw_size, H, D, C, T, t0, Tl, precision, qhat,
                              kvhat, qproj, kproj, minmax, stream, 3, nullptr, 0);
}

// internal (common.h): K as in hept_prep_hash_rpe
int hept_prep_hash_fused_rpe(const float* x, const float* norm_w, const float* norm_b, float eps, const float* w_q,
                             const float* w_k, const float* w_v, const float* coords, const float* sqrt_w, int K,
                             const float* alpha, const int64_t* codes, int N, int raw_size, int H, int D, int C, int T,
                             int t0, int Tl, int precision, void* qhat, void* kvhat, float* qproj, float* kproj,
                             float* minmax, void* stream, void* zero_ptr, size_t zero_bytes) {
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    if (zero_bytes % 4 != 0 || (zero_bytes && !zero_ptr) || zero_bytes > 0xFFFFFFFFull) return HEPT_ERR_ARG;
    const ZeroJob zero{reinterpret_cast<unsigned int*>(zero_ptr), (unsigned int)(zero_bytes / 4)};
    if (K < 0 || (K > 0 && (C < 2 || H * (C - 1) * K > 1024))) return HEPT_ERR_SHAPE;
    if (!x || !norm_w || !norm_b || !w_q || !w_k || !w_v || !coords || !sqrt_w || !alpha || !qhat || !kvhat ||
        !qproj || !kproj || !minmax)
        return HEPT_ERR_ARG;
    if (H != 8 || D != 24 || N < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (precision == HEPT_PREC_F32_MFMA || precision == HEPT_PREC_F32_DIFF) precision = HEPT_PREC_F32;  // same f32 tile rows, another block_attn kernel
    if (precision != HEPT_PREC_F32 && precision != HEPT_PREC_BF16 && precision != HEPT_PREC_MIXED16)
        return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define HEPT_FUSED_CASE(CC)                                                                                         \
    if (C == CC)                                                                                                    \
        return launch_prep_fused<CC>(x, norm_w, norm_b, eps, w_q, w_k, w_v, coords, sqrt_w, K, alpha, codes, N, raw_size, \
                                     T, t0, Tl, precision, qhat, kvhat, qproj, kproj, minmax, st, zero);
    HEPT_FUSED_CASE(6)
    HEPT_FUSED_CASE(4)
    HEPT_FUSED_CASE(2)
#undef HEPT_FUSED_CASE
    return HEPT_ERR_SHAPE;
}

extern "C" int hept_prep_hash_fused(const float* x, const float* norm_w, const float* norm_b, float eps,
                                    const float* w_q, const float* w_k, const float* w_v, const float* coords,
                                    const float* sqrt_w, const float* alpha, const int64_t* codes, int N, int raw_size,
                                    int H, int D, int C, int T, int t0, int Tl, int precision, void* qhat,
                                    void* kvhat, float* qproj, float* kproj, float* minmax, void* stream) {
    return hept_prep_hash_fused_rpe(x, norm_w, norm_b, eps, w_q, w_k, w_v, coords, sqrt_w, 0, alpha, codes, N, raw_size, H,
                                    D, C, T, t0, Tl, precision, qhat, kvhat, qproj, kproj, minmax, stream, nullptr, 0);
}
