// block_attn: gather -> block-local RBF attention on MFMA -> scatter to original order.
//
// Replaces, per (table, head, block) (reference file:line):
//   sort_to_buckets x3 / batched_index_select   example/hept.py:70-72, example/hept_utils.py:74-92
//   qkv_res                                      example/hept.py:7-18
//   invert_permutation + unsort_from_buckets x2  example/hept.py:76-78, example/hept_utils.py:50-61,95-97
//
// One workgroup = one block of B sorted queries x the B sorted keys of the same rank range
// (queries and keys are sorted independently, as in the reference).  NKT = ceil(B/32) waves;
// wave w owns queries [32w, 32w+32) and walks the NKT 32-key tiles:
//
//   X  = K^ . Q^T      (keys on rows -> accumulator registers, queries on lanes)
//        accumulator initialised with (-0.5|q|^2) + (-0.5|k|^2), so X is the full logit
//   P  = exp(min(X,0)) in registers; the accumulator layout of X is already the A-operand
//        layout of the next product (k index = key), no LDS bounce, no shuffles
//   Z += P^T-as-A . V  where V carries a 1.0 column at index D: Z[:, D] is the row sum
//
// K^ and V tiles are staged once per block in LDS (rows gathered by kpos, 16 B per lane,
// one kvhat row = K^ row | V row, contiguous); the wave's 32 Q^ rows go straight from HBM to
// registers in B-operand layout.  No row max is needed: logits are clamped to <= 0
// (example/hept.py:12) and the reference combines un-normalised numerators/denominators.
// Output rows are written straight to part[t][qpos[row]][h][:], which fuses the un-sort:
//   fp32 path: 32 floats = one 128-B line  [numer 0..D-1 | denom at D | 0]
//   bf16 path: 16 dwords = 64 B            [numer as D bf16 in dwords 0..11 | denom f32 in dword 12 | 0]
//
// bf16 path: v_mfma_f32_32x32x16_bf16, V fragments by ds_read_b64_tr_b16 (mixed16: the K^.Q^T product
// takes fp16 rows through v_mfma_f32_32x32x16_f16; P.V stays bf16 for the exponent range of tiny weights).
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), plain ds_read_b32 for V.
// blockIdx % H = head: with H = 8 every XCD (round-robin dispatch) gathers from one head's
// qhat/kvhat slab only (speed only; nothing depends on placement).
#include "common.h"
#include "p2p_dev.h"

namespace {

// Cache policy of the packed partial rows: 1 = non-temporal (they are written once and read by the
// combine only: as plain stores they push the gathered rows out of the L2), 0 = plain (A/B).  The f32 rows stay plain:
// the f32 combine lives on what the caches still hold of them.
#ifndef HEPT_ATTN_STORE
#define HEPT_ATTN_STORE 1
#endif
__device__ __forceinline__ void store16_rows(unsigned int* dst, const u32x4& v) {
#if HEPT_ATTN_STORE == 1
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));   // (sc1 / sc0 sc1 stores gain nothing: r06_experiments.txt)
#else
    *reinterpret_cast<u32x4*>(dst) = v;
#endif
}

// Register budget: a workgroup lives ~10 us, most of it waiting for its gathered rows, so throughput is set by how
// many workgroups a CU holds.  16-bit tiles: 8 waves per SIMD (<= 64 VGPRs, no spills; 87 -> 80 us at tracking-60k);
// f32 tiles carry twice the fragments: 6 waves per SIMD, 4 for their ragged-tile variants (masks in registers).  The
// ragged 16-bit variants fit 64 VGPRs as well (B = 100, the reference's yaml: 97 -> see DESIGN.md section 6).
// (ragged 16-bit tiles -- B = 100, the reference's yaml -- carry their masks in registers: at the 64-VGPR cap of 8 waves
//  per SIMD they spilled 1-5 registers to scratch; 7 waves = 72 VGPRs, no scratch: 193.3 -> 191.1 us per forward)
#ifndef HEPT_RAGGED16_WAVES
#define HEPT_RAGGED16_WAVES 7
#endif
// (B = 96 with f32 partial rows -- D != 24, an off-the-shipped-shapes instance -- spilled 4 registers at the 64-VGPR cap as
//  well, and at 72: 6 waves = 80 VGPRs)
constexpr int attn_waves(int nkt, bool bf16, bool p16, bool full, bool diff = false) {
    if (diff) return full ? 3 : 2;   // the difference form keeps 16 more accumulators: 157 / 180 VGPRs, no scratch (an accuracy mode)
    return bf16 ? ((nkt == 3 && !p16 && full) ? 6 : (full ? 8 : HEPT_RAGGED16_WAVES)) : (full ? 6 : 4);
}
// DIFF (HEPT_PREC_F32_DIFF, f32 rows on the native f32 MFMA): the logit -|q^ - k^|^2 / 2 with the COORDINATE part as explicit
// differences.  The reference (example/hept.py:8-12) forms q^.k^ - |q^|^2/2 - |k^|^2/2; with the shipped checkpoint's
// sqrt_w (up to 5.8e3) on raw coordinates those three terms are ~3e8 each and their fp32 sum is rounding noise (sigma ~ 20
// in the logit, in the reference's own evaluation as well: DESIGN.md section 4, case G7).  Here only the D feature columns
// go through the matrix product (q.k - |q|^2/2 - |k|^2/2 over d < D: O(10) terms), and every column from D on (the C
// coordinate columns sqrt_w . coords and the zero padding up to column 29) contributes -(q^_c - k^_c)^2 / 2 from the
// difference of the two stored values: no cancellation.  Mathematically the same logit; numerically the accurate one.
template <int NKT, bool BF16, bool P16, bool F16QK, bool FULL, bool DIFF = false>
__global__ __launch_bounds__(64 * NKT) __attribute__((amdgpu_waves_per_eu(attn_waves(NKT, BF16, P16, FULL, DIFF), attn_waves(NKT, BF16, P16, FULL, DIFF))))
void block_attn_kernel(const char* __restrict__ qhat,
                                                              const char* __restrict__ kvhat,
                                                              const int* __restrict__ qpos,
                                                              const int* __restrict__ kpos,
                                                              float* __restrict__ part, int N, int H, int D, int B,
                                                              int nb, HeadRange hr, PushArgs pa) {
    // one-sided table sharding: the first pa.push_wgs workgroups of the launch do not compute attention but sum and
    // send the rows of the PREVIOUS head group (p2p_dev.h); they sit at the front of the grid so that they start
    // with the kernel and run beside the attention workgroups for its whole length
    if (pa.push_wgs > 0 && (int)blockIdx.x < pa.push_wgs) {
        reduce_push_body<P16>(pa, (int)blockIdx.x);
        return;
    }
    constexpr int NT = 64 * NKT;
    constexpr int KEYS = 32 * NKT;
    constexpr int ESZ = BF16 ? 2 : 4;
    constexpr int QROW = 32 * ESZ;    // bytes of a q^ (or k^, or v) row
    constexpr int KVROW = 2 * QROW;   // k^ row | v row
    constexpr int CH = QROW / 16;     // 16-B chunks per k^ (or v) row: 4 / 8
    constexpr int CPR = 2 * CH;       // chunks per kvhat row

    // all LDS in one dynamic array (fp32 tiles at B=256 need 66 KB > the 64 KB static limit)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_s = smem;
    char* v_s = smem + KEYS * QROW;
    float* kn_s = reinterpret_cast<float*>(smem + 2 * KEYS * QROW);
    int* qidx_s = reinterpret_cast<int*>(smem + 2 * KEYS * QROW + KEYS * 4);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int bid = (int)blockIdx.x - pa.push_wgs;
    const int h = hr.h0 + bid % hr.hg;
    const int rest = bid / hr.hg;
    const int b = hr.tl > 0 ? rest / hr.tl : rest % nb, t = hr.tl > 0 ? rest % hr.tl : rest / nb;
    const size_t seg = ((size_t)t * H + h) * N + (size_t)b * B;
    const int* __restrict__ kp = kpos + seg;
    const int* __restrict__ qp = qpos + seg;
    const char* __restrict__ qbase = qhat + (size_t)h * N * QROW;
    const char* __restrict__ kvbase = kvhat + (size_t)h * N * KVROW;

    // ---- this wave's 32 query rows: HBM -> registers (B-operand layout), norm from the row tail
    const int qi = w * 32 + li;
    const bool qvalid = FULL || qi < B;
    const int qsrc = qp[qvalid ? qi : 0];
    if (hh == 0) qidx_s[qi] = qvalid ? qsrc : -1;
    const char* qrow = qbase + (size_t)qsrc * QROW;
    float qn = *reinterpret_cast<const float*>(qrow + QROW - 4);
    u32x4 qraw[BF16 ? 2 : 4];
#pragma unroll
    for (int s = 0; s < (BF16 ? 2 : 4); ++s)
        qraw[s] = *reinterpret_cast<const u32x4*>(qrow + (BF16 ? (s * 32 + hh * 16) : (hh * 64 + s * 16)));
    if (hh == 1) qraw[BF16 ? 1 : 3][3] = 0u;  // the norm slot is not a feature
    if constexpr (DIFF) {
        static_assert(!BF16, "the difference form runs on f32 rows");
        // this lane holds columns 16 hh + 4 s + j: feature norm over the columns below D, everything from D on leaves
        // the matrix product
        float part = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool feat = 16 * hh + 4 * s2 + j < D;
                const float qv = __uint_as_float(qraw[s2][j]);
                part = feat ? fmaf(qv, qv, part) : part;
                if (!feat) qraw[s2][j] = 0u;
            }
        qn = -0.5f * (part + __shfl_xor(part, 32));
    }

    // ---- stage K^ and V tiles: gathered rows, 16 B per lane, K^ XOR-swizzled against bank conflicts
#pragma unroll
    for (int it = 0; it < CPR / 2; ++it) {
        const int ci = it * NT + tid;
        const int key = ci / CPR, c = ci % CPR;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (FULL || key < B) {  // FULL: B == 32 * NKT, no ragged tile -> no masking code at all
            const int src = kp[key];   // (the gathers and the positions WANT the caches: nt loads are slower, r06_experiments.txt)
            val = *reinterpret_cast<const u32x4*>(kvbase + (size_t)src * KVROW + c * 16);
        }
        if (c < CH) {
            if (c == CH - 1) {
                kn_s[key] = __uint_as_float(val[3]);
                val[3] = 0u;
            }
            const int sw = BF16 ? ((key >> 2) & 3) : ((key >> 1) & 7);
            *reinterpret_cast<u32x4*>(k_s + key * QROW + ((c ^ sw) * 16)) = val;
        } else {
            *reinterpret_cast<u32x4*>(v_s + key * QROW + (c - CH) * 16) = val;
        }
    }
    __syncthreads();
    if constexpr (DIFF) {
        // -|k|^2 / 2 over the feature columns of every staged key (the stored norm covers the coordinate columns too)
        if (tid < KEYS) {
            const int sw = (tid >> 1) & 7;
            float acc = 0.f;
            for (int col = 0; col < D; ++col) {
                const float kv2 = *reinterpret_cast<const float*>(k_s + tid * QROW + (((col >> 2) ^ sw) * 16) + (col & 3) * 4);
                acc = fmaf(kv2, kv2, acc);
            }
            kn_s[tid] = -0.5f * acc;
        }
        __syncthreads();
    }

    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;

#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (!FULL && kt * 32 >= B) break;  // uniform
        const int key = kt * 32 + li;
        f32x16 x;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = qn + kn_s[kt * 32 + hept_acc_row(r, hh)];

        if constexpr (BF16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c = 2 * s + hh;
                const u32x4 kraw =
                    *reinterpret_cast<const u32x4*>(k_s + key * QROW + ((c ^ ((key >> 2) & 3)) * 16));
                if constexpr (F16QK)
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kraw),
                                                               __builtin_bit_cast(f16x8, qraw[s]), x, 0, 0, 0);
                else
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kraw),
                                                                __builtin_bit_cast(bf16x8, qraw[s]), x, 0, 0, 0);
            }
        } else {
            float kf[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(
                    k_s + key * QROW + (((4 * hh + c) ^ ((key >> 1) & 7)) * 16));
                kf[4 * c] = kk[0]; kf[4 * c + 1] = kk[1]; kf[4 * c + 2] = kk[2]; kf[4 * c + 3] = kk[3];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s)
                x = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], __uint_as_float(qraw[s >> 2][s & 3]), x, 0, 0, 0);
            if constexpr (DIFF) {
                // columns D .. 29: -(q^_c - k^_c)^2 / 2 from the stored values (the query's from its row -- an L1 hit --,
                // the keys' from the staged tile: the lanes of a half read one key, a broadcast)
                float dsq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dsq[r] = 0.f;
                for (int col = D; col < 30; ++col) {
                    const float qc = *reinterpret_cast<const float*>(qrow + col * 4);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kr = kt * 32 + hept_acc_row(r, hh);
                        const float kc = *reinterpret_cast<const float*>(
                            k_s + kr * QROW + (((col >> 2) ^ ((kr >> 1) & 7)) * 16) + (col & 3) * 4);
                        const float dlt = qc - kc;
                        dsq[r] = fmaf(dlt, dlt, dsq[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = fmaf(-0.5f, dsq[r], x[r]);
            }
        }

        // exp(min(x, 0)) written as min(exp(x), 1): identical value for every x (exp is monotone, exp(0) = 1),
        // and v_min on the v_exp result needs no NaN-canonicalising v_max in front of it
        float pr[16];
        exp_clamped(x, pr);
        if (!FULL && (kt + 1) * 32 > B) {  // ragged last tile (B not a multiple of 32): padded keys carry no weight
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + hept_acc_row(r, hh) >= B) pr[r] = 0.f;
        }

        if constexpr (BF16) {
            const int g = lane >> 4, l16 = lane & 15;
            const int vrow = kt * 32 + 4 * hh + (l16 >> 2);
            const int vcol = 32 * (g & 1) + 8 * (l16 & 3);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 pw;
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j] = hept_pack_bf16(pr[8 * s + 2 * j], pr[8 * s + 2 * j + 1]);
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4_ptr)(v_s + (vrow + 16 * s) * QROW + vcol));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4_ptr)(v_s + (vrow + 16 * s + 8) * QROW + vcol));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pw),
                                                            __builtin_bit_cast(bf16x8, vv), z, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vv =
                    *reinterpret_cast<const float*>(v_s + (kt * 32 + hept_acc_row(r, hh)) * QROW + li * 4);
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], vv, z, 0, 0, 0);
            }
        }
    }

    if constexpr (P16) {
        // ---- scatter, 64-B rows [24 bf16 numer | f32 denom | 0].  The wave's 32 x 32 accumulator tile goes to LDS as it
        //      stands (the K^ / V tiles are dead after the loop: f32, one ds_write_b32 per register at an immediate offset),
        //      and the lane that stores piece pc of row `row` reads its eight values back (two ds_read_b128) and packs them
        //      there: a lane stores a 16-B piece, four lanes a whole 64-B row -- two store instructions per wave, and the
        //      form the xGMI links want when the row belongs to another rank (direct mode).  (Round 5 packed every register
        //      where it lay: a DPP move, a conversion, two selects and a swizzled address per register -- 16 x 6 vector
        //      instructions of a kernel that issues one in 57 % of its cycles: 56.2 -> 51.8 us.)  Word i = pack(column 2i,
        //      column 2i + 1): the same values and rounding as ever.
        __syncthreads();   // every wave is done reading k_s / v_s
        float* tile = reinterpret_cast<float*>(smem) + w * 32 * 32;   // this wave's 4 KiB of the dead K^ / V tiles
        {
            float* wr = tile + (4 * hh) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * 32] = z[r];
        }
        unsigned int* __restrict__ pt =
            reinterpret_cast<unsigned int*>(part) + (size_t)t * hr.tstride_rows * 16 + (size_t)(h - hr.hsub) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (lane >> 2) + 16 * j, pc = lane & 3;
            const int q2 = w * 32 + row;
            const f32x4 f0 = *reinterpret_cast<const f32x4*>(tile + row * 32 + pc * 8);
            const f32x4 f1 = *reinterpret_cast<const f32x4*>(tile + row * 32 + pc * 8 + 4);
            u32x4 v = {hept_pack_bf16(f0[0], f0[1]), hept_pack_bf16(f0[2], f0[3]), hept_pack_bf16(f1[0], f1[1]),
                       hept_pack_bf16(f1[2], f1[3])};
            // P16 rows exist for D == 24 only: piece 3 = [denominator + 1e-20 (example/hept.py:14) as f32 | 0 | 0 | 0]
            if (pc == 3) v = u32x4{__float_as_uint(f0[0] + 1e-20f), 0u, 0u, 0u};
            if (FULL || q2 < B) {
                const int dst = qidx_s[q2];
                if (pa.direct) {
                    bool remote;
                    char* rowp = direct_row(pa, dst, h - hr.h0, 64, remote) + pc * 16;
                    if (remote) store16_system(rowp, v);
                    else *reinterpret_cast<u32x4*>(rowp) = v;
                } else {
                    store16_rows(pt + (size_t)dst * hr.hout * 16 + pc * 4, v);
                }
            }
        }
    } else {
        // ---- scatter: row = 32 floats = one 128-B line per query, lanes 0..31 contiguous
        float* __restrict__ pt = part + (size_t)t * hr.tstride_rows * 32 + (size_t)(h - hr.hsub) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q2 = w * 32 + hept_acc_row(r, hh);
            if (FULL || q2 < B) {
                const int dst = qidx_s[q2];
                float val = z[r];
                if (li == D) val += 1e-20f;  // example/hept.py:14
                if (pa.direct) {   // (f32 rows to another rank: 4-byte system-scope stores -- correct, not tuned)
                    bool remote;
                    char* rowp = direct_row(pa, dst, h - hr.h0, 128, remote) + li * 4;
                    if (remote) store4_system(rowp, __float_as_uint(val));
                    else *reinterpret_cast<float*>(rowp) = val;
                } else {
                    pt[(size_t)dst * hr.hout * 32] = val;
                }
            }
        }
    }
    // (direct mode: this rank's row flags are raised by the next kernel of the stream, p2p_dev.h raise_flags)
    if (pa.direct) {
        drain_remote_stores();
        if (pa.counter) signal_when_all_done(pa.counter, pa.peers, pa.world, pa.flag_idx, pa.epoch, gridDim.x - pa.push_wgs);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// f32 tiles on the bf16 matrix pipe ("split" kernel, the default for HEPT_PREC_F32).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and the f32-tile kernel above is bound by it (49 % MFMA
// busy at 300 us).  A product of two f32 values is recovered to f32 accuracy from bf16 pieces: a = ah + am + al
// with ah = bf16(a), am = bf16(a - ah), al = bf16(a - ah - am) (each residual is exact in f32), and
//     a.b = ah.bh + (ah.bm + am.bh) + (am.bm + ah.bl + al.bh) + O(2^-26 |a||b|)
// -- six bf16 MFMAs with f32 accumulation instead of eight f32 MFMAs of 1/16 the rate each (3/8 of the issue
// slots, 1/16 of the cycles per slot).  The logits K^.Q^ use all six terms (they are differences of large numbers); V
// keeps VP = 3 planes (with VP = 2 the output error is 7x larger); P, a weight in [0, 1], is split into HEPT_SPLIT_PP
// pieces -- two since round 4, see below: the accuracy contract of HEPT_PREC_F32 is "f32 rows and accumulation, the
// logits to f32 accuracy, P to 16 significand bits"; HEPT_PREC_F32_MFMA is the exact f32 mode.  K^ and V are split
// once per workgroup while they are staged (3 + VP bf16 planes in LDS, 384 B per key instead of 256; 64 keys at a
// time, the next 64 gathered rows wait in registers while the current ones are computed), the wave's Q^ rows once
// into registers, P in registers after the exp.  The norms -|q|^2/2, -|k|^2/2 ride in the product
// through the two spare columns (30, 31) against a 1.0 on the other side.
// Inputs, outputs and the (t, block, head) -> workgroup map are those of the f32-tile kernel.
// (Round 3, the reference's own block_size 100 = three full 32-row tiles + 4 rows: both kernels were also built on 16-row
//  tiles, v_mfma_f32_16x16x32_bf16 -- 7 x 7 tile pairs instead of 4 x 4 padded ones.  16-bit tiles: 90.2 us against 89.9 us
//  (bound by the gathers, not by issue).  f32 split tiles, two query tiles per wave: 197 us against 213 us as a kernel
//  timed back to back on cache-warm rows, but 357 us against 355 us per forward in place, where its gathers run cold --
//  the one-tile-per-wave form was LDS-bound at 241 us.  Neither was kept: DESIGN.md section 6.)
// Pieces of P in the P.V product.  V keeps three planes (every value to 24 bits); P = exp(.) in [0, 1] takes TWO:
// ph + pm carries 16 significand bits, |P - ph - pm| <= 2^-18 P, unbiased, so the product terms are
// ph.(vh + vm + vl) + pm.(vh + vm) -- 5 MFMAs instead of 6 and 3 VALU instructions per weight instead of 5.5.  The
// output is a weighted mean of value rows: its error from this is <= 3.8e-6 max|v| in the worst case (one dominant key)
// and ~1/sqrt(keys) of that typically; measured against the three-piece build at tracking-60k: largest element
// difference 3.6e-6 = 0.23x of the stated fp32 tolerance (atol 1e-5 + rtol 1e-4), kernel 167.6 -> 158.7 us.
// Keys staged per chunk.  Every chunk costs two workgroup barriers, and between them a wave alternates a staging stretch
// (VALU: the splits) and a compute stretch (MFMA + the P splits); at B = 256 (8 waves, 2 workgroups per CU) the two
// waves a workgroup keeps on a SIMD move in lockstep, so half of the SIMD's chances to run one wave's MFMAs beside
// another's VALU are gone -- longer stretches (128 keys: 2 chunks instead of 4) give the other workgroup's waves more
// room to fall out of step.  The planes are then 48 KB per workgroup, which 2 workgroups per CU can afford.
#ifndef HEPT_SPLIT_CK8
#define HEPT_SPLIT_CK8 128
#endif
constexpr int split_ck(int nkt) { return nkt == 8 ? HEPT_SPLIT_CK8 : (nkt >= 2 ? 64 : 32); }   // (128 keys at B = 128: slower, r06_experiments.txt)
#ifndef HEPT_SPLIT_PP
#define HEPT_SPLIT_PP 2
#endif
// DIFF (HEPT_PREC_F32_DIFF at D = 24): the difference form of block_attn_kernel<.., DIFF> on this kernel -- the 24 feature
// columns through the six-term split products, the columns from 24 on as -(q^_c - k^_c)^2 / 2 from the stored f32 values
// (the keys' six coordinate columns in a side array of the staged chunk, the feature-only norms formed while staging).
template <int NKT, bool FULL, int VP, bool DIFF = false>
__global__ __launch_bounds__(64 * NKT) void block_attn_split_kernel(const float* __restrict__ qhat,
                                                                    const float* __restrict__ kvhat,
                                                                    const int* __restrict__ qpos,
                                                                    const int* __restrict__ kpos,
                                                                    float* __restrict__ part, int N, int H, int D,
                                                                    int B, int nb, HeadRange hr, PushArgs pa) {
    if (pa.push_wgs > 0 && (int)blockIdx.x < pa.push_wgs) {   // see block_attn_kernel
        reduce_push_body<false>(pa, (int)blockIdx.x);
        return;
    }
    constexpr int NT = 64 * NKT;
    constexpr int KEYS = 32 * NKT;
    constexpr int PROW = 64;                    // bytes of one 32-column bf16 plane row
    constexpr int CK = split_ck(NKT);           // keys staged at a time: 24 KB of planes (48 KB at B = 256)
    constexpr int NCH = (KEYS + CK - 1) / CK;   // chunks
    constexpr int IPT = (CK * 8 + NT - 1) / NT; // 8-column items per thread and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_s = smem;                   // 3 planes [CK][32 bf16], 16-B chunks XOR-swizzled
    char* v_s = smem + 3 * CK * PROW;   // VP planes [CK][32 bf16], read transposed
    float* kc_s = reinterpret_cast<float*>(smem + (3 + VP) * CK * PROW);   // DIFF: [CK][8] columns 24 .. 29 of the chunk's keys, f32

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int bid = (int)blockIdx.x - pa.push_wgs;
    const int h = hr.h0 + bid % hr.hg;
    const int rest = bid / hr.hg;
    const int b = hr.tl > 0 ? rest / hr.tl : rest % nb, t = hr.tl > 0 ? rest % hr.tl : rest / nb;
    const size_t seg = ((size_t)t * H + h) * N + (size_t)b * B;
    const int* __restrict__ kp = kpos + seg;
    const int* __restrict__ qp = qpos + seg;
    const float* __restrict__ qbase = qhat + (size_t)h * N * 32;
    const float* __restrict__ kvbase = kvhat + (size_t)h * N * 64;

    // ---- gathered kvhat rows of chunk ch -> registers (one item = 8 consecutive columns; c < 4: K^, c >= 4: V).
    //      The loads of chunk ch + 1 are issued before chunk ch is computed, so their latency hides behind it.
    float pre[IPT][8];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const int ci = it * NT + tid;
            const int key = ch * CK + (ci >> 3), c = ci & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) pre[it][j] = 0.f;
            if (ci < CK * 8 && key < (FULL ? KEYS : B)) {
                const int ks = kp[key];
                if (c >= 4 && hr.vsrc) {
                    // value columns 8 (c - 4) .. + 7 of [v (D) | 1.0 | 0 ..] straight from the caller's v (D % 4 == 0:
                    // whole 16-B pieces); padding rows of the src variant (>= raw_size) are zero but for the 1.0
                    const int col = 8 * (c - 4);
                    const float* src = hr.vsrc + ((size_t)ks * H + h) * D + col;
                    const bool real = ks < hr.raw_size;
                    if (real && col < D) {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
                        pre[it][0] = a0[0]; pre[it][1] = a0[1]; pre[it][2] = a0[2]; pre[it][3] = a0[3];
                    }
                    if (real && col + 4 < D) {
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + 4);
                        pre[it][4] = a1[0]; pre[it][5] = a1[1]; pre[it][6] = a1[2]; pre[it][7] = a1[3];
                    }
                    if (col == D) pre[it][0] = 1.f;
                    if (col + 4 == D) pre[it][4] = 1.f;
                } else {
                    const float* src = kvbase + (size_t)ks * 64 + c * 8;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + 4);
                    pre[it][0] = a0[0]; pre[it][1] = a0[1]; pre[it][2] = a0[2]; pre[it][3] = a0[3];
                    pre[it][4] = a1[0]; pre[it][5] = a1[1]; pre[it][6] = a1[2]; pre[it][7] = a1[3];
                    if (c == 3) pre[it][6] = 1.f;  // k^ columns (30, 31) = (1, -|k|^2/2)
                }
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const int ci = it * NT + tid;
            if (ci >= CK * 8) break;
            const int row = ci >> 3, c = ci & 7;
            u32x4 ph, pm, pl;
            if constexpr (DIFF) {
                // the eight items of a key are eight consecutive lanes: items 0 .. 2 hold its 24 feature columns, item 3
                // columns 24 .. 31.  Feature norm: three partial sums meet in item 3's lane (DPP row_shr); its six
                // coordinate columns go to the side array and leave the product, column 31 takes -|k_feat|^2 / 2
                float sq = 0.f;
                if (c < 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sq = fmaf(pre[it][j], pre[it][j], sq);
                }
                const int sqi = __builtin_bit_cast(int, sq);
                const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, sqi, 0x111, 0xF, 0xF, true));
                const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, sqi, 0x112, 0xF, 0xF, true));
                const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, sqi, 0x113, 0xF, 0xF, true));
                if (c == 3) {
                    *reinterpret_cast<f32x4*>(kc_s + row * 8) = f32x4{pre[it][0], pre[it][1], pre[it][2], pre[it][3]};
                    *reinterpret_cast<f32x4*>(kc_s + row * 8 + 4) = f32x4{pre[it][4], pre[it][5], 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 6; ++j) pre[it][j] = 0.f;
                    pre[it][7] = -0.5f * ((s3 + s2) + s1);
                }
            }
            if (c < 4) {
                split3_bf16(pre[it], ph, pm, pl);
                const int off = row * PROW + ((c ^ ((row >> 2) & 3)) * 16);
                *reinterpret_cast<u32x4*>(k_s + off) = ph;
                *reinterpret_cast<u32x4*>(k_s + CK * PROW + off) = pm;
                *reinterpret_cast<u32x4*>(k_s + 2 * CK * PROW + off) = pl;
            } else {
                if constexpr (VP == 3) split3_bf16(pre[it], ph, pm, pl);
                else split2_bf16(pre[it], ph, pm);
                const int off = row * PROW + (c - 4) * 16;
                *reinterpret_cast<u32x4*>(v_s + off) = ph;
                *reinterpret_cast<u32x4*>(v_s + CK * PROW + off) = pm;
                if constexpr (VP == 3) *reinterpret_cast<u32x4*>(v_s + 2 * CK * PROW + off) = pl;
            }
        }
    };
    fetch(0);

    // ---- this wave's 32 query rows -> three bf16 planes in registers (B-operand layout: lane-half hh of step s
    //      holds columns 16 s + 8 hh .. + 8)
    const int qi = w * 32 + li;
    const bool qvalid = FULL || qi < B;
    const int qsrc = qp[qvalid ? qi : 0];
    const float* qrow = qbase + (size_t)qsrc * 32;
    u32x4 qh[2], qm[2], ql[2];
    float qc[DIFF ? 6 : 1];   // DIFF: this query's columns 24 .. 29 (both lane halves: the logits of a query live in both)
    {
        float av[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hh);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hh + 4);
            av[s][0] = a0[0]; av[s][1] = a0[1]; av[s][2] = a0[2]; av[s][3] = a0[3];
            av[s][4] = a1[0]; av[s][5] = a1[1]; av[s][6] = a1[2]; av[s][7] = a1[3];
        }
        if constexpr (DIFF) {
            // lane half 0 holds columns 0 .. 7 and 16 .. 23, half 1 columns 8 .. 15 and 24 .. 31: the feature norm is the sum
            // of both halves' squares, the coordinates come from half 1
            float fn = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) fn = fmaf(av[0][j], av[0][j], fn);
            if (hh == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) fn = fmaf(av[1][j], av[1][j], fn);
            }
            fn += __shfl_xor(fn, 32);
#pragma unroll
            for (int c = 0; c < 6; ++c) qc[c] = __shfl(av[1][c], li + 32);
            if (hh == 1) {
#pragma unroll
                for (int j = 0; j < 6; ++j) av[1][j] = 0.f;
                av[1][6] = -0.5f * fn;   // meets k^[30] = 1
                av[1][7] = 1.f;          // meets k^[31] = -|k_feat|^2 / 2
            }
        } else if (hh == 1) {  // the norms ride in the product: q^[30] = -|q|^2/2 meets k^[30] = 1, q^[31] = 1
            av[1][6] = av[1][7];   // meets k^[31] = -|k|^2/2 (columns E..30 of the stored rows are zero, E <= 30)
            av[1][7] = 1.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) split3_bf16(av[s], qh[s], qm[s], ql[s]);
    }

    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    // lane part of the K^ chunk addresses (tile bases are multiples of 32 rows: the swizzle term is the lane's)
    const int krow[2] = {li * PROW + ((hh ^ ((li >> 2) & 3)) * 16), li * PROW + (((2 + hh) ^ ((li >> 2) & 3)) * 16)};
    const int vlane = (4 * hh + ((lane & 15) >> 2)) * PROW + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (!FULL && ch * CK >= B) break;  // uniform
        if (ch > 0) __syncthreads();       // every wave is done with the previous chunk's planes
        stage();
        if (ch + 1 < NCH) fetch(ch + 1);
        __syncthreads();
        // Q^.K^ of BOTH tiles of the chunk first: the second tile's MFMA chain has no dependence on the first tile's exp /
        // split VALU work, so the two can share the SIMD (one wave's instruction stream is in order: a P tile computed
        // right behind its own logits leaves the matrix pipe idle for the whole exp / split stretch)
        constexpr int TPC = CK / 32, GRP = TPC < 2 ? TPC : 2;   // tiles per chunk; tiles whose logits are formed together
#pragma unroll
        for (int g0 = 0; g0 < TPC; g0 += GRP) {
        f32x16 xs[GRP];
#pragma unroll
        for (int kg = 0; kg < GRP; ++kg) {
            const int kl = g0 + kg;
            const int kt = ch * TPC + kl;
            if (kt >= NKT) break;
            if (!FULL && kt * 32 >= B) break;  // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[kg][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = kl * 32 * PROW + krow[s];
                const u32x4 kh = *reinterpret_cast<const u32x4*>(k_s + off);
                const u32x4 km = *reinterpret_cast<const u32x4*>(k_s + CK * PROW + off);
                const u32x4 kl3 = *reinterpret_cast<const u32x4*>(k_s + 2 * CK * PROW + off);
                xs[kg] = mfma_bf16(kl3, qh[s], xs[kg]);
                xs[kg] = mfma_bf16(kh, ql[s], xs[kg]);
                xs[kg] = mfma_bf16(km, qm[s], xs[kg]);
                xs[kg] = mfma_bf16(km, qh[s], xs[kg]);
                xs[kg] = mfma_bf16(kh, qm[s], xs[kg]);
                xs[kg] = mfma_bf16(kh, qh[s], xs[kg]);
            }
        }
#pragma unroll
        for (int kg = 0; kg < GRP; ++kg) {
            const int kl = g0 + kg;
            const int kt = ch * TPC + kl;
            if (kt >= NKT) break;
            if (!FULL && kt * 32 >= B) break;  // uniform
            f32x16 x = xs[kg];
            if constexpr (DIFF) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* kc = kc_s + (kl * 32 + hept_acc_row(r, hh)) * 8;   // (a broadcast: one key per lane half)
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(kc), k1 = *reinterpret_cast<const f32x4*>(kc + 4);
                    float d0 = qc[0] - k0[0], d1 = qc[1] - k0[1], d2 = qc[2] - k0[2], d3 = qc[3] - k0[3];
                    float d4 = qc[4] - k1[0], d5 = qc[5] - k1[1];
                    float dsq = d0 * d0;
                    dsq = fmaf(d1, d1, dsq); dsq = fmaf(d2, d2, dsq); dsq = fmaf(d3, d3, dsq);
                    dsq = fmaf(d4, d4, dsq); dsq = fmaf(d5, d5, dsq);
                    x[r] = fmaf(-0.5f, dsq, x[r]);
                }
            }

            float pr[16];
            exp_clamped(x, pr);
            if (!FULL && (kt + 1) * 32 > B) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + hept_acc_row(r, hh) >= B) pr[r] = 0.f;
            }

#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float pa[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[j] = pr[8 * s + j];
                constexpr int PP = HEPT_SPLIT_PP;   // pieces of P (the V planes are VP)
                u32x4 pp[3];
                if constexpr (PP == 3) split3_bf16(pa, pp[0], pp[1], pp[2]);
                else split2_bf16(pa, pp[0], pp[1]);
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                u32x4 vpl[VP];
#pragma unroll
                for (int pl = 0; pl < VP; ++pl) {
                    const char* vb = v_s + pl * CK * PROW + (kl * 32 + 16 * s) * PROW + vlane;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vb));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vb + 8 * PROW));
                    const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    vpl[pl] = __builtin_bit_cast(u32x4, vv);
                }
                // every product p_i . v_j of order i + j <= 2, smallest first
#pragma unroll
                for (int ord = 2; ord >= 0; --ord)
#pragma unroll
                    for (int i = PP - 1; i >= 0; --i) {
                        const int j = ord - i;
                        if (j >= 0 && j < VP) z = mfma_bf16(pp[i], vpl[j], z);
                    }
            }
        }
        }   // tile groups of the chunk
    }

    // ---- scatter: row = 32 floats = one 128-B line per query.  Round 6: the wave's 32 x 32 accumulator tile passes through
    //      LDS (the planes are dead: f32, one ds_write_b32 per register at an immediate offset), and a lane stores a 16-B
    //      piece, eight lanes a whole row: four store instructions per wave with four address computations instead of
    //      sixteen 4-byte ones with a shuffle and a 64-bit multiply each (~110 vector instructions of the wave's ~870).
    __syncthreads();   // every wave is done with the last chunk's planes
    float* tile = reinterpret_cast<float*>(smem) + w * 32 * 32;
    {
        float* wr = tile + (4 * hh) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * 32] = z[r];
    }
    float* __restrict__ pt = part + (size_t)t * hr.tstride_rows * 32 + (size_t)(h - hr.hsub) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (lane >> 3) + 8 * j, pc = lane & 7;
        const int q2 = w * 32 + row;
        const int dst = __shfl(qsrc, row);   // lane li holds the source row of query 32 w + li
        f32x4 f = *reinterpret_cast<const f32x4*>(tile + row * 32 + pc * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pc * 4 + e == D) f[e] += 1e-20f;  // example/hept.py:14
        if (FULL || q2 < B) {
            if (pa.direct) {   // see block_attn_kernel
                bool remote;
                char* rowp = direct_row(pa, dst, h - hr.h0, 128, remote) + pc * 16;
                if (remote) store16_system(rowp, __builtin_bit_cast(u32x4, f));
                else *reinterpret_cast<f32x4*>(rowp) = f;
            } else {
                *reinterpret_cast<f32x4*>(pt + (size_t)dst * hr.hout * 32 + pc * 4) = f;
            }
        }
    }
    if (pa.direct) {   // see block_attn_kernel
        drain_remote_stores();
        if (pa.counter) signal_when_all_done(pa.counter, pa.peers, pa.world, pa.flag_idx, pa.epoch, gridDim.x - pa.push_wgs);
    }
}

template <bool FULL, int VP, bool DIFF = false>
int launch_attn_split(int nkt, dim3 grid, hipStream_t st, const float* qhat, const float* kvhat, const int* qpos,
                      const int* kpos, float* part, int N, int H, int D, int B, int nb, HeadRange hr, PushArgs pa) {
#define HEPT_SPLIT_CASE(K)                                                                                       \
    case K: {                                                                                                    \
        /* the planes, or the epilogue's 4 KiB per wave where that is more (B = 224: 7 waves) */                  \
        constexpr size_t lds0 = (size_t)(3 + VP) * split_ck(K) * 64 + (DIFF ? (size_t)split_ck(K) * 32 : 0);        \
        constexpr size_t lds = lds0 > (size_t)K * 4096 ? lds0 : (size_t)K * 4096;                                \
        if (lds > 65536) {                                                                                       \
            static LdsRaised raised;                                                                             \
            if (hept_raise_lds(raised, reinterpret_cast<const void*>(&block_attn_split_kernel<K, FULL, VP, DIFF>), lds)) \
                return HEPT_ERR_LAUNCH;                                                                          \
        }                                                                                                        \
        hipLaunchKernelGGL((block_attn_split_kernel<K, FULL, VP, DIFF>), grid, dim3(64 * K), lds, st, qhat, kvhat, qpos, \
                           kpos, part, N, H, D, B, nb, hr, pa);                                                        \
        break;                                                                                                   \
    }
    switch (nkt) {
        HEPT_SPLIT_CASE(1)
        HEPT_SPLIT_CASE(2)
        HEPT_SPLIT_CASE(3)
        HEPT_SPLIT_CASE(4)
        HEPT_SPLIT_CASE(5)
        HEPT_SPLIT_CASE(6)
        HEPT_SPLIT_CASE(7)
        HEPT_SPLIT_CASE(8)
        default:
            return HEPT_ERR_SHAPE;
    }
#undef HEPT_SPLIT_CASE
    return hept_launch_status();
}

template <bool BF16, bool P16, bool F16QK, bool FULL, bool DIFF = false>
int launch_attn_full(int nkt, dim3 grid, hipStream_t st, const char* qhat, const char* kvhat, const int* qpos,
                const int* kpos, float* part, int N, int H, int D, int B, int nb, HeadRange hr, PushArgs pa) {
#define HEPT_ATTN_CASE(K)                                                                                    \
    case K: {                                                                                                \
        constexpr size_t lds = (size_t)2 * 32 * K * 32 * (BF16 ? 2 : 4) + 32 * K * 8;                        \
        if (lds > 65536) {                                                                                   \
            static LdsRaised raised;                                                                         \
            if (hept_raise_lds(raised, reinterpret_cast<const void*>(&block_attn_kernel<K, BF16, P16, F16QK, FULL, DIFF>), lds)) \
                return HEPT_ERR_LAUNCH;                                                                      \
        }                                                                                                    \
        hipLaunchKernelGGL((block_attn_kernel<K, BF16, P16, F16QK, FULL, DIFF>), grid, dim3(64 * K), lds, st, qhat, kvhat, qpos, \
                           kpos, part, N, H, D, B, nb, hr, pa);                                                    \
        break;                                                                                               \
    }
    switch (nkt) {
        HEPT_ATTN_CASE(1)
        HEPT_ATTN_CASE(2)
        HEPT_ATTN_CASE(3)
        HEPT_ATTN_CASE(4)
        HEPT_ATTN_CASE(5)
        HEPT_ATTN_CASE(6)
        HEPT_ATTN_CASE(7)
        HEPT_ATTN_CASE(8)
        default:
            return HEPT_ERR_SHAPE;
    }
#undef HEPT_ATTN_CASE
    return hept_launch_status();
}

template <bool BF16, bool P16, bool F16QK, bool DIFF = false>
int launch_attn(int nkt, dim3 grid, hipStream_t st, const char* qhat, const char* kvhat, const int* qpos,
                const int* kpos, float* part, int N, int H, int D, int B, int nb, HeadRange hr, PushArgs pa) {
    if (B == 32 * nkt)
        return launch_attn_full<BF16, P16, F16QK, true, DIFF>(nkt, grid, st, qhat, kvhat, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    return launch_attn_full<BF16, P16, F16QK, false, DIFF>(nkt, grid, st, qhat, kvhat, qpos, kpos, part, N, H, D, B, nb, hr, pa);
}

}  // namespace

namespace {
int block_attn_impl(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos, int N, int H, int D,
                    int Tl, int B, int precision, const HeadRange& hr_in, float* part, void* stream,
                    const PushArgs* push = nullptr, VSrc vs = VSrc{}) {
    HeadRange hr = hr_in;
    hr.vsrc = nullptr;
    hr.raw_size = N;
    static const bool table_major = [] { const char* e = getenv("HEPT_ATTN_TABLE_MAJOR"); return e && *e && *e != '0'; }();
    hr.tl = table_major ? 0 : Tl;
    if (vs.v) {   // only the f32-row split kernel reads v in place (D % 4 == 0: 16-B pieces of the caller's rows)
        if (precision != HEPT_PREC_F32 || D % 4 != 0 || (reinterpret_cast<uintptr_t>(vs.v) & 15)) return HEPT_ERR_ARG;
        hr.vsrc = vs.v;
        hr.raw_size = vs.raw_size;
    }
    if (!qhat || !kvhat || !qpos || !kpos || (!part && !(push && push->direct))) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || B < 1 || B > HEPT_MAX_BLOCK || N % B != 0 || D < 1 || D > 28)
        return HEPT_ERR_SHAPE;
    if (hr.h0 < 0 || hr.hg < 1 || hr.h0 + hr.hg > H || hr.hout < 1 || hr.hsub < 0 || hr.hsub > hr.h0 ||
        hr.h0 + hr.hg - hr.hsub > hr.hout)
        return HEPT_ERR_SHAPE;

    const int nb = N / B, nkt = (B + 31) / 32;
    PushArgs pa{};
    if (push) {
        pa = *push;
        // carried push: the rows being sent must be in the format this launch's kernels write (the previous group's
        // launch wrote them); direct scatter: no pushing workgroups, one local table, the launch's heads = the group
        if (pa.direct ? (pa.push_wgs != 0 || Tl != 1 || pa.h0 != hr.h0 || pa.hg != hr.hg) : pa.push_wgs < 1)
            return HEPT_ERR_ARG;
    }
    const dim3 grid((unsigned)((size_t)Tl * nb * hr.hg + pa.push_wgs));
    hipStream_t st = (hipStream_t)stream;
    // bf16 tiles with D == 24 write packed 64-B partial rows (HEPT_PART_PACKED), everything else 128-B f32 rows
    const char* qh = (const char*)qhat;
    const char* kv = (const char*)kvhat;
    if (precision == HEPT_PREC_BF16 && D == 24)
        return launch_attn<true, true, false>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    if (precision == HEPT_PREC_BF16)
        return launch_attn<true, false, false>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    if (precision == HEPT_PREC_MIXED16 && D == 24)
        return launch_attn<true, true, true>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    if (precision == HEPT_PREC_MIXED16)
        return launch_attn<true, false, true>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    if (precision == HEPT_PREC_F32) {
        const float* qf = (const float*)qhat;
        const float* kf = (const float*)kvhat;
        if (B == 32 * nkt) return launch_attn_split<true, 3>(nkt, grid, st, qf, kf, qpos, kpos, part, N, H, D, B, nb, hr, pa);
        return launch_attn_split<false, 3>(nkt, grid, st, qf, kf, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    }
    if (precision == HEPT_PREC_F32_MFMA)
        return launch_attn<false, false, false>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    if (precision == HEPT_PREC_F32_DIFF) {
        // D = 24 (every shipped shape): the difference form on the split-bf16 kernel (1.3x the "fp32" time instead of 2x);
        // other head dimensions: on the f32-MFMA kernel (HEPT_DIFF_MFMA=1: that kernel at D = 24 too, for A/B)
        const char* e = getenv("HEPT_DIFF_MFMA");   // (read on every call: the parity tests switch it inside one process)
        const bool mfma = e && *e && *e != '0';
        if (D == 24 && !mfma) {
            const float* qf = (const float*)qhat;
            const float* kf = (const float*)kvhat;
            if (B == 32 * nkt) return launch_attn_split<true, 3, true>(nkt, grid, st, qf, kf, qpos, kpos, part, N, H, D, B, nb, hr, pa);
            return launch_attn_split<false, 3, true>(nkt, grid, st, qf, kf, qpos, kpos, part, N, H, D, B, nb, hr, pa);
        }
        return launch_attn<false, false, false, true>(nkt, grid, st, qh, kv, qpos, kpos, part, N, H, D, B, nb, hr, pa);
    }
    return HEPT_ERR_SHAPE;
}
}  // namespace

// internal: hept_block_attn_heads whose launch also carries the one-sided push of another head group (comm.h)
int hept_block_attn_heads_push(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos, int N,
                               int H, int D, int Tl, int B, int precision, int h0, int hg, int hout, int hsub,
                               int n_rows_out, float* part, const PushArgs* push, void* stream, VSrc vs) {
    if (n_rows_out < N) return HEPT_ERR_SHAPE;
    const HeadRange hr{h0, hg, hout, hsub, (long long)n_rows_out * hout, nullptr, 0, 0};
    return block_attn_impl(qhat, kvhat, qpos, kpos, N, H, D, Tl, B, precision, hr, part, stream, push, vs);
}

extern "C" int hept_block_attn(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos, int N,
                               int H, int D, int Tl, int B, int precision, float* part, void* stream) {
    const HeadRange all{0, H, H, 0, (long long)N * H, nullptr, 0, 0};
    return block_attn_impl(qhat, kvhat, qpos, kpos, N, H, D, Tl, B, precision, all, part, stream);
}

extern "C" int hept_block_attn_heads(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos,
                                     int N, int H, int D, int Tl, int B, int precision, int h0, int hg, int hout,
                                     int hsub, int n_rows_out, float* part, void* stream) {
    if (n_rows_out < N) return HEPT_ERR_SHAPE;
    const HeadRange hr{h0, hg, hout, hsub, (long long)n_rows_out * hout, nullptr, 0, 0};
    return block_attn_impl(qhat, kvhat, qpos, kpos, N, H, D, Tl, B, precision, hr, part, stream);
}
