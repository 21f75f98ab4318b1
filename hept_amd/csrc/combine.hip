// combine_out: OR-combine over tables + output projection; reduce_tables for table sharding.
//
// Replaces (reference file:line):
//   out = o.sum(0) / logits.sum(0)                         example/hept.py:79
//   out_linear(rearrange(out, "h n d -> n (h d)"))         example/hept.py:80
//
// HBM-bound: reads Tl*H partial rows per point (part layout (Tl, N, H, row): the H rows of a point
// are one contiguous run per table; row = 128 B of f32, or the packed 64-B form the bf16 path of
// block_attn writes), writes D floats per point.  One wave owns 32
// points: lane (point = lane & 31, half = lane >> 5) loads 64 contiguous bytes of each
// (table, head) row with 16-B loads, sums the tables in the reference's order, divides by the
// denominator column and feeds the 32 x (H*32) tile straight to v_mfma_f32_32x32x2_f32 against
// the LDS-resident, zero-padded transpose of out_linear.weight (exact fp32 fma chain).
#include "common.h"
#include "p2p_dev.h"

namespace {

#ifndef HEPT_CMB_STG_AUX
#define HEPT_CMB_STG_AUX 2   // cache policy bits of the direct-to-LDS row loads (2 = nt: the partial rows are read once)
#endif
constexpr int CMB_THREADS = 256;
constexpr int CMB_WAVES = CMB_THREADS / HEPT_WAVE;
constexpr int WT_PITCH = 33;  // LDS pitch of one (head, d) weight row: odd, so the transposing writes of the staging
                              // loop (consecutive d) land on consecutive banks (pitch 32: 32-way conflicts)

// Raw 16-B pieces of one (table, point, head) partial row kept in registers until they are needed
// (P16: 4 pieces = 64 B; f32: 7 pieces = the 28 leading floats, which hold numer 0..D-1 and the denominator).
template <bool P16>
struct RawRow {
    static constexpr int PIECES = P16 ? 4 : 7;
    u32x4 q[PIECES];
    __device__ __forceinline__ void load(const float* __restrict__ row) {
        const u32x4* src = reinterpret_cast<const u32x4*>(row);
#pragma unroll
#ifdef HEPT_CMB_NT_LOADS
        for (int i = 0; i < PIECES; ++i) q[i] = __builtin_nontemporal_load(src + i);
#else
        for (int i = 0; i < PIECES; ++i) q[i] = src[i];
#endif
    }
    // x[0..27] += widened row; returns the denominator
    __device__ __forceinline__ float add_to(float (&x)[28], int D) const {
        if constexpr (P16) {  // [24 bf16 numerators in dwords 0..11 | f32 denominator in dword 12 | 0]
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[8 * p + 2 * i] += hept_bf16_lo(q[p][i]);
                    x[8 * p + 2 * i + 1] += hept_bf16_hi(q[p][i]);
                }
            return __uint_as_float(q[3][0]);
        } else {              // [numer 0..D-1 | denom at D | 0]
            float den = 0.f;
#pragma unroll
            for (int p = 0; p < 7; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = __uint_as_float(q[p][u]);
                    x[4 * p + u] += v;
                    if (4 * p + u == D) den = v;
                }
            return den;
        }
    }
};

// One wave = 32 points.  Lane (point = lane & 31, hh = lane >> 5) owns the rows of heads 2j + hh: it loads a
// whole row (64 B packed / 112 B f32, contiguous) per table, so the denominator sits in the same lane as its
// numerators (no cross-lane traffic), and the two lane halves feed the two k-slots of v_mfma_f32_32x32x2_f32
// with different heads: D steps per head pair, no padded columns in the MFMA stream.  The rows of the next
// head pair are requested before the current pair is reduced (the kernel runs at ~2 waves per SIMD, so the
// overlap has to come from inside the wave).  Tables are summed in the reference's order (t = 0, 1, ...);
// the division is one correctly rounded reciprocal per row followed by multiplies (<= 1 ulp from a/b).
// FFN = true (SURVEY.md §8 f-4, D = 24): the rest of the Attn block in the epilogue (example/transformer.py:161-165,
// eval mode: dropout is the identity): y1 = x + aggr; y = y1 + ff.2(relu(ff.0(norm2(y1)))).  The 32 x D tile of
// aggregated rows goes through a wave-private LDS tile so that lane (point, half) holds a whole row; each half
// computes 12 hidden and 12 output units (weights are LDS broadcasts), the halves swap hidden units by shuffle.
struct FfnIn {
    const float* x;      // (n_count, D) block input rows, row n0 first
    const float* ln_w;   // norm2.weight, norm2.bias
    const float* ln_b;
    const float* w1;     // ff.0.weight (D, D), ff.0.bias
    const float* b1;
    const float* w2;     // ff.2.weight (D, D), ff.2.bias
    const float* b2;
    float eps;
};
constexpr int FFN_D = 24, FFN_PITCH = 25, FFN_WFLOATS = 2 * FFN_D * FFN_D + 4 * FFN_D;

// DT = compile-time head dimension (0: use the runtime value).  With a runtime D the `u < D` guards of the MFMA loop
// became one branch + one fully exposed LDS round trip per step (ds_read, s_waitcnt lgkmcnt(0), v_mfma, 96 times
// per wave); with DT the loop is straight-line code.
// SPLIT (few tiles: a 6k cloud, or the N/G points a table-sharded rank finishes): the four waves of a workgroup
// share ONE tile, each reduces every fourth head pair, and the partial 32 x D products meet in LDS -- a quarter of
// the serial chain per wave when there are not enough tiles to fill the machine anyway.
// PUSH (one-sided table sharding, D = 24, p2p.hip): `part` is this rank's receive region -- the kernel first waits for
// the arrival flags of every (head group, source rank) -- and the finished rows go to this rank's slice of the
// gathered output in EVERY rank's exchange buffer: the 32 x D tile passes through LDS so that each lane stores 16-B
// pieces of contiguous rows (a 4-byte scatter per accumulator register would put 4-byte writes on the xGMI links);
// the kernel that follows on the stream (the gather) raises this rank's output flag everywhere.
// STG (D = 24, an even number of heads per head group -- the plain (Tl, N, H, row) layout and, since round 5, the receive
// region of the table-sharded step, whose rows lie in uncached memory where a 16-B piece per lane is a memory request of
// its own): the rows reach the lanes THROUGH LDS.  A lane that
// loads "its" row touches 64 different lines per wave instruction, 16 B of each (the texture addresser works through them
// one by one, and the L1 has to hold every line until its fourth piece is asked for); here a wave instruction is a
// direct-to-LDS load (global_load_lds_dwordx4) of 8 points x 128 B -- the two rows of a head pair, whole lines -- into a
// wave-private image [table][point][8 pieces], no registers, and lane (point, half) then reads its 4 pieces with
// ds_read_b128 (pieces XOR-permuted by the point on the SOURCE side, so that 16 lanes hit 16 bank groups).  The next
// head pair's loads are issued as soon as the current rows are in registers: the prefetch lives in LDS (12 KB per
// wave), not in a second register set.
constexpr int STG_TABLE_BYTES = 32 * 8 * 16;             // packed rows: one table, one head pair, 32 points
constexpr int STG_WAVE_BYTES = 3 * STG_TABLE_BYTES;      // up to three tables in flight
// f32 rows (128 B): the image of ONE (table, head pair) unit is 32 points x 256 B = 8 KB, so the tables of a head pair
// go through a two-deep ring of such images one after the other: unit u + 2 is requested as soon as unit u's rows are
// in registers, the wait in front of unit u leaves the 8 loads of unit u + 1 in flight (s_waitcnt vmcnt(8)), and the
// weight column comes straight from W (no slab: 64 KB of images per workgroup, two workgroups per CU).
constexpr int STG32_UNIT_BYTES = 32 * 16 * 16;
constexpr int STG32_WAVE_BYTES = 2 * STG32_UNIT_BYTES;
template <bool P16, bool FFN = false, int DT = 0, bool SPLIT = false, bool PUSH = false, bool STG = false>
__global__ __launch_bounds__(CMB_THREADS) void combine_out_kernel(const float* __restrict__ part, int Tl, int N,
                                                                  int H, int D_rt, int n0, int n_count,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out, int HG, size_t gstride,
                                                                  FfnIn ffn = FfnIn{}, P2pDev px = P2pDev{},
                                                                  int stg_off = 0) {
    static_assert(!PUSH || (DT == 24 && !FFN), "the pushing epilogue is built for D = 24 rows");
    static_assert(!STG || DT == 24, "staged rows: D = 24");
    constexpr int ROWF = P16 ? 16 : 32;   // row pitch in 4-byte units
    extern __shared__ __attribute__((aligned(16))) float wt_s[];  // [H (even-padded)][28 (d)][32 (c, zero padded)]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int D = DT ? DT : D_rt;
    const int HD = H * D, HP = (H + 1) & ~1;
    // W is (D, H*D) row-major: read it coalesced and transpose while writing LDS (a transposed gather from global
    // cost 6.6 us per launch).  Slab of head h: [d][c] at pitch WT_PITCH, only the D x D entries that are read are
    // written (lanes >= D read column D-1: their MFMA output columns are never stored); the pad head of an odd H is 0.
    // Packed rows (16-bit tiles): the weight column of a lane comes straight from W -- 24 consecutive floats of row
    // `lc`, six 16-B loads per head pair that hit L1 / L2 (W is 18 KB) -- so the launch has no staging prologue and no
    // barrier in front of its first tile: 27.3 -> 25.6 us at tracking-60k.  f32 rows loaded lane by lane keep the LDS slab
    // (their waves already carry 21 row loads per head pair, and six more cost 4.5 us: 54.7 against 50.2); STAGED f32
    // rows (their row loads are direct-to-LDS, no registers) take the column from W as well.  Both forms read W and the
    // rows as 16-B pieces: combine_launch refuses bases that are not 16-B aligned.
    constexpr bool WDIRECT = (P16 || STG) && DT == 24;
    if constexpr (!WDIRECT) {
        for (int i = tid; i < D * HD; i += CMB_THREADS) {
            const int c = i / HD, h = (i % HD) / D, d = i % D;
            wt_s[(h * 28 + d) * WT_PITCH + c] = W[i];
        }
        if (HP > H)
            for (int i = tid; i < 28 * WT_PITCH; i += CMB_THREADS) wt_s[H * 28 * WT_PITCH + i] = 0.f;
    }
    const int lc = li < D ? li : D - 1;
    const float bia = (li < D && bias) ? bias[li] : 0.f;
    float* ffn_s = wt_s + (WDIRECT ? 0 : HP * 28 * WT_PITCH);     // [w1 | w2 | b1 | b2 | ln_w | ln_b] (no slab with WDIRECT)
    float* stage_s = ffn_s + FFN_WFLOATS + w * 32 * FFN_PITCH;  // this wave's 32 x D tile
    float* stage_s_end = ffn_s + (FFN ? FFN_WFLOATS + CMB_WAVES * 32 * FFN_PITCH : 0);
    if constexpr (PUSH) {
        // the launches that stored this rank's rows into the owners' buffers have ended: announce them (raise_flags)
        if (px.consumer_raises && blockIdx.x < RAISE_WGS)
            raise_flags(px.peers, px.world, px.wait_groups, px.me, HEPT_MAX_RANKS_DEV, px.epoch);
        if (tid < px.wait_groups * px.world) {
            const int g = tid / px.world, src = tid - g * px.world;
            wait_flag(flag_word(px.local, g * HEPT_MAX_RANKS_DEV + src), px.epoch, px.status, 1u, px.timeout);
        }
        __syncthreads();   // nobody reads a received row before every flag has been seen
    }
    // a wait that timed out (now or in an earlier step: the status word is sticky) means some received rows may be
    // unfinished: this rank's slice then goes out as NaN, so that no rank can take it for a result
    bool poisoned = false;
    if constexpr (PUSH) poisoned = status_bad(px.status);
    // The first tile's rows are requested BEFORE the weights are staged: every wave of the launch is resident at once
    // (one tile per wave at tracking-60k), so without this the whole chip spends the staging prologue (~3 us) with
    // no row in flight.
    const size_t tstride = (size_t)N * HG * ROWF;
    const int n_tiles = (n_count + 31) / 32;
    const int hp0 = SPLIT ? 2 * w : 0, hstep = SPLIT ? 2 * CMB_WAVES : 2;
    const int tpre = Tl < 3 ? Tl : 3;
    // table t's rows.  PUSH: the "tables" are the source ranks' slices of the receive region, and the slice this rank
    // sent to itself lies in ordinary (cached) device memory instead (P2pDev::self_rows, same layout)
    auto tbl = [&](int t) {
        if constexpr (PUSH) return (t == px.me ? px.self_rows : part) + (size_t)t * tstride;
        else return part + (size_t)t * tstride;
    };
    auto row_at = [&](int tile, int hp) {   // offset of this lane's row inside a table
        const int i = tile * 32 + li;
        const int n = n0 + (i < n_count ? i : n_count - 1);
        const int head = hp + hh < H ? hp + hh : 0, g = head / HG;
        return (size_t)n * HG * ROWF + (size_t)g * gstride + (size_t)(head - g * HG) * ROWF;
    };
    const int tile_first = SPLIT ? blockIdx.x : blockIdx.x * CMB_WAVES + w;
    RawRow<P16> cur[3], nxt[3];  // up to 3 tables in flight; more tables are loaded in place below
    // STG: this wave's LDS image and the two halves of the hand-over (see the comment above the kernel)
    char* const stg = reinterpret_cast<char*>(wt_s) + stg_off + w * (P16 ? STG_WAVE_BYTES : STG32_WAVE_BYTES);
    // ---- f32 rows: the unit ring (see STG32_UNIT_BYTES)
    struct Unit { int tile, hp, t; bool ok; };
    const int tile_step = SPLIT ? (int)gridDim.x : (int)gridDim.x * CMB_WAVES;
    auto unit_next = [&](Unit u) {
        if (!u.ok) return u;
        if (u.t + 1 < Tl) { ++u.t; return u; }
        u.t = 0;
        if (u.hp + hstep < HP) { u.hp += hstep; return u; }
        u.hp = hp0;
        u.tile += tile_step;
        u.ok = u.tile < n_tiles;
        return u;
    };
    auto unit_request = [&](const Unit& u, int slot_buf) {   // 8 direct-to-LDS loads: 32 points x (2 rows x 8 pieces)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int slot = j * 64 + lane, pt = slot >> 4, sub = (slot & 15) ^ (pt & 15);
            const int i2 = u.tile * 32 + pt;
            const int n = n0 + (i2 < n_count ? i2 : n_count - 1);
            const int grp = u.hp / HG;   // (a head pair never straddles two head groups: HG is even)
            const char* g = reinterpret_cast<const char*>(tbl(u.t) + (size_t)grp * gstride) +
                            ((size_t)n * HG + (u.hp - grp * HG)) * 128 + sub * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(stg + slot_buf * STG32_UNIT_BYTES + j * 1024),
                                             16, 0, HEPT_CMB_STG_AUX);
        }
    };
    int ring = 0;
    auto stg_request = [&](int tile, int hp) {   // rows of head pair (hp, hp + 1), tables 0 .. tpre - 1: global -> LDS
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < tpre) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int slot = j * 64 + lane, pt = slot >> 3, sub = (slot & 7) ^ ((pt >> 1) & 7);
                    const int i2 = tile * 32 + pt;
                    const int n = n0 + (i2 < n_count ? i2 : n_count - 1);
                    const int grp = hp / HG;
                    const char* g = reinterpret_cast<const char*>(tbl(t) + (size_t)grp * gstride) +
                                    ((size_t)n * HG + (hp - grp * HG)) * 64 + sub * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(stg + t * STG_TABLE_BYTES + j * 1024),
                                                     16, 0, HEPT_CMB_STG_AUX);
                }
            }
    };
    auto stg_take = [&]() {   // LDS -> this lane's rows (cur[t]); afterwards the image may be overwritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < tpre) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cur[t].q[q] = *reinterpret_cast<const u32x4*>(stg + t * STG_TABLE_BYTES +
                                                                   (li * 8 + ((hh * 4 + q) ^ ((li >> 1) & 7))) * 16);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (tile_first < n_tiles && hp0 < HP) {
        if constexpr (STG && !P16) {
            const Unit u0{tile_first, hp0, 0, true}, u1 = unit_next(u0);
            unit_request(u0, 0);
            if (u1.ok) unit_request(u1, 1);
        } else if constexpr (STG) stg_request(tile_first, hp0);
        else {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < tpre) cur[t].load(tbl(t) + row_at(tile_first, hp0));
        }
    }
    if constexpr (FFN) {
        for (int i = tid; i < FFN_D * FFN_D; i += CMB_THREADS) {
            ffn_s[i] = ffn.w1[i];
            ffn_s[FFN_D * FFN_D + i] = ffn.w2[i];
        }
        if (tid < FFN_D) {
            ffn_s[2 * FFN_D * FFN_D + tid] = ffn.b1[tid];
            ffn_s[2 * FFN_D * FFN_D + FFN_D + tid] = ffn.b2[tid];
            ffn_s[2 * FFN_D * FFN_D + 2 * FFN_D + tid] = ffn.ln_w[tid];
            ffn_s[2 * FFN_D * FFN_D + 3 * FFN_D + tid] = ffn.ln_b[tid];
        }
    }
    __syncthreads();
    // rows of head group g = head / HG live in their own (Tl, N, HG, row) buffer at part + g * gstride (table
    // sharding receives the head groups one exchange at a time); HG == H: the plain (Tl, N, H, row) layout
    float* red_s = stage_s_end;  // SPLIT: [CMB_WAVES - 1][16][64] partial accumulators of waves 1..
    // PUSH: this wave's 32 x 24 tile as a contiguous image of the output rows (wave 0 only under SPLIT)
    float* push_s = red_s + (SPLIT ? (CMB_WAVES - 1) * 16 * 64 : 0) + (SPLIT ? 0 : w * 32 * 24);
    for (int tile = tile_first; tile < n_tiles; tile += SPLIT ? gridDim.x : gridDim.x * CMB_WAVES) {
        const int i = tile * 32 + li;
        auto row_of = [&](int hp) { return row_at(tile, hp); };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // (the first tile's rows were requested before the weight staging; a wave that takes several tiles requests the
        //  next tile's first rows while it finishes the current one)
        const int tile_next = tile + (SPLIT ? (int)gridDim.x : (int)gridDim.x * CMB_WAVES);
        for (int hp = hp0; hp < HP; hp += hstep) {
            const bool more = hp + hstep < HP;
            const bool wrap = !more && tile_next < n_tiles;   // the last head pair of a tile that is not the wave's last
            // packed rows: the next head pair's rows are requested before the current pair is unpacked.  f32 rows are
            // 21 x 16 B per head pair and lane: holding two sets put the kernel at 256 VGPRs + 120 AGPRs, ONE wave per
            // SIMD and therefore two rounds of workgroups; there the current set is summed first and the next set is
            // loaded into the same registers, still ahead of the 24 MFMAs (see below)
            if constexpr (STG && P16) {
                stg_take();
                if (more) stg_request(tile, hp + hstep);
                else if (wrap) stg_request(tile_next, hp0);
            } else if constexpr (P16) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if ((more || wrap) && t < tpre)
                        nxt[t].load(tbl(t) + (more ? row_of(hp + hstep) : row_at(tile_next, hp0)));
            }
            // this head's weight column, requested before the rows are unpacked (LDS latency hides under the VALU work)
            float wv[28];
            if constexpr (WDIRECT) {
                const float* wrow = W + (size_t)lc * HD + (size_t)(hp + hh < H ? hp + hh : 0) * D;
#pragma unroll
                for (int u4 = 0; u4 < 7; ++u4)
                    if (4 * u4 < D) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + 4 * u4);
                        wv[4 * u4] = w4[0]; wv[4 * u4 + 1] = w4[1]; wv[4 * u4 + 2] = w4[2]; wv[4 * u4 + 3] = w4[3];
                    }
                if (hp + hh >= H) {   // the pad head of an odd H
#pragma unroll
                    for (int u = 0; u < 28; ++u) wv[u] = 0.f;
                }
            } else {
                const float* wrow = wt_s + (size_t)(hp + hh) * 28 * WT_PITCH + lc;
#pragma unroll
                for (int u = 0; u < 28; ++u)
                    if (u < D) wv[u] = wrow[u * WT_PITCH];
            }
            float s[28];
#pragma unroll
            for (int u = 0; u < 28; ++u) s[u] = 0.f;
            float den = 0.f;
            if constexpr (STG && !P16) {
                for (int t = 0; t < Tl; ++t) {   // the tables of this head pair, one ring slot each
                    const Unit ua{tile, hp, t, true}, ub = unit_next(ua), uc = unit_next(ub);
                    // unit ua's loads were issued before ub's eight: leave those in flight (nothing else is younger
                    // than them but loads that may as well be waited for)
                    if (ub.ok) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 7; ++q)
                        cur[0].q[q] = *reinterpret_cast<const u32x4*>(stg + ring * STG32_UNIT_BYTES +
                                                                       (li * 16 + ((hh * 8 + q) ^ (li & 15))) * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (uc.ok) unit_request(uc, ring);
                    den += cur[0].add_to(s, D);
                    ring ^= 1;
                }
            } else {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < tpre) den += cur[t].add_to(s, D);
            for (int t = 3; t < Tl; ++t) {  // beyond three tables: plain loads
                RawRow<P16> extra;
                extra.load(tbl(t) + row_of(hp));
                den += extra.add_to(s, D);
            }
            }
            if constexpr (!P16 && !STG) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if ((more || wrap) && t < tpre)
                        cur[t].load(tbl(t) + (more ? row_of(hp + hstep) : row_at(tile_next, hp0)));
            }
            const float inv = 1.0f / den;
#pragma unroll
            for (int u = 0; u < 28; ++u)
                if (u < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[u] * inv, wv[u], acc, 0, 0, 0);
            if constexpr (P16 && !STG) {
#pragma unroll
                for (int t = 0; t < 3; ++t) cur[t] = nxt[t];
            }
        }
        if constexpr (SPLIT) {
            if (w != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red_s[((w - 1) * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (w != 0) {
                __syncthreads();  // pairs with the barrier after wave 0's epilogue (red_s is reused by the next tile)
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float tot = acc[r];
#pragma unroll
                for (int ww = 1; ww < CMB_WAVES; ++ww) tot += red_s[((ww - 1) * 16 + r) * 64 + lane];
                acc[r] = tot;
            }
        }
        if constexpr (FFN) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (li < FFN_D) stage_s[hept_acc_row(r, hh) * FFN_PITCH + li] = acc[r] + bia;
            // (one wave's LDS accesses execute in order: the tile is complete when the reads below are issued)
            const bool valid = i < n_count;
            const f32x4* xs = reinterpret_cast<const f32x4*>(ffn.x + (size_t)(valid ? i : n_count - 1) * FFN_D);
            float y1[FFN_D];
#pragma unroll
            for (int j = 0; j < FFN_D / 4; ++j) {
                const f32x4 xv = xs[j];
#pragma unroll
                for (int u = 0; u < 4; ++u) y1[4 * j + u] = xv[u] + stage_s[li * FFN_PITCH + 4 * j + u];
            }
            float mean = 0.f;
#pragma unroll
            for (int j = 0; j < FFN_D; ++j) mean += y1[j];
            mean *= 1.0f / FFN_D;
            float var = 0.f;
#pragma unroll
            for (int j = 0; j < FFN_D; ++j) { const float dlt = y1[j] - mean; var = fmaf(dlt, dlt, var); }
            const float rstd = 1.0f / sqrtf(var * (1.0f / FFN_D) + ffn.eps);
            const float* g_s = ffn_s + 2 * FFN_D * FFN_D + 2 * FFN_D;
            float z[FFN_D];
#pragma unroll
            for (int j = 0; j < FFN_D; ++j) z[j] = (y1[j] - mean) * rstd * g_s[j] + g_s[FFN_D + j];
            // hidden units [12 hh, 12 hh + 12) of ff.0 + ReLU
            float hid[FFN_D];
#pragma unroll
            for (int uu = 0; uu < FFN_D / 2; ++uu) {
                const int u = hh * (FFN_D / 2) + uu;
                const float* wr = ffn_s + u * FFN_D;
                float a = ffn_s[2 * FFN_D * FFN_D + u];
#pragma unroll
                for (int j = 0; j < FFN_D; ++j) a = fmaf(wr[j], z[j], a);
                hid[uu] = fmaxf(a, 0.f);
            }
            // swap halves: afterwards hid[0..11] = units 0..11, hid[12..23] = units 12..23 in both lanes
#pragma unroll
            for (int uu = 0; uu < FFN_D / 2; ++uu) {
                const float other = __shfl_xor(hid[uu], 32);
                hid[FFN_D / 2 + uu] = hh ? hid[uu] : other;
                hid[uu] = hh ? other : hid[uu];
            }
            float yo[FFN_D / 2];
#pragma unroll
            for (int uu = 0; uu < FFN_D / 2; ++uu) {
                const int u = hh * (FFN_D / 2) + uu;
                const float* wr = ffn_s + FFN_D * FFN_D + u * FFN_D;
                float a = ffn_s[2 * FFN_D * FFN_D + FFN_D + u];
#pragma unroll
                for (int j = 0; j < FFN_D; ++j) a = fmaf(wr[j], hid[j], a);
                yo[uu] = a;
            }
            if (valid) {
                f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)i * FFN_D + hh * (FFN_D / 2));
#pragma unroll
                for (int c4 = 0; c4 < FFN_D / 8; ++c4) {
                    f32x4 o;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // y1[12 hh + 4 c4 + u] with a compile-time index in both halves
                        const float res = hh ? y1[FFN_D / 2 + 4 * c4 + u] : y1[4 * c4 + u];
                        o[u] = res + yo[4 * c4 + u];
                    }
                    dst[c4] = o;
                }
            }
        } else if constexpr (PUSH) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (li < 24) push_s[hept_acc_row(r, hh) * 24 + li] = acc[r] + bia;
            // (one wave's LDS accesses execute in order: the tile is complete when the reads below are issued)
            const int rows = n_count - tile * 32 < 32 ? n_count - tile * 32 : 32;
            const size_t tile_off = px.slice_off + (size_t)tile * 32 * 24 * 4;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int c = lane + 64 * kk;            // 16-B piece c of the tile's 192: floats [4c, 4c + 4), row c / 6
                if (c < rows * 6) {
                    u32x4 v = *reinterpret_cast<const u32x4*>(push_s + 4 * c);
                    if (poisoned) v = u32x4{0x7FC00000u, 0x7FC00000u, 0x7FC00000u, 0x7FC00000u};
                    for (int s = 0; s < px.world; ++s) {
                        // (own slice: an ordinary store into the caller's output, when the step has one -- the gather
                        //  then copies the other ranks' slices only; view mode reads it in the exchange buffer)
                        if (s == px.me && px.out_local)
                            *reinterpret_cast<u32x4*>(px.out_local + tile_off - px.out_base + (size_t)c * 16) = v;
                        else store16_system(px.peers[s] + tile_off + (size_t)c * 16, v);
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i2 = tile * 32 + hept_acc_row(r, hh);
                if (i2 < n_count && li < D) hept_st<HEPT_NT_OUT>(out + (size_t)i2 * D + li, acc[r] + bia);
            }
        }
        if constexpr (SPLIT) __syncthreads();
    }
    // (PUSH: this rank's output flag is raised by the gather kernel that follows, p2p.hip)
    if constexpr (PUSH) {
        drain_remote_stores();
        if (px.out_counter) signal_when_all_done(px.out_counter, px.peers, px.world, OUT_FLAG_WORD + px.me, px.epoch, gridDim.x);
    }
}

// 16 consecutive columns [16*hh, 16*hh+16) of one partial row, widened to fp32 (reduce_tables)
template <bool P16>
__device__ __forceinline__ void load_half_row(const float* __restrict__ row, int hh, float (&x)[16]) {
    if constexpr (P16) {
        const u32x4* src = reinterpret_cast<const u32x4*>(row) + 2 * hh;
        const u32x4 a = src[0], b = src[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = hept_bf16_lo(a[i]);
            x[2 * i + 1] = hept_bf16_hi(a[i]);
            x[8 + 2 * i] = hept_bf16_lo(b[i]);
            x[8 + 2 * i + 1] = hept_bf16_hi(b[i]);
        }
        if (hh == 1) {  // dword 12 (= b[0]) is the f32 denominator -> column 24; the rest is padding
            x[8] = __uint_as_float(b[0]);
#pragma unroll
            for (int i = 9; i < 16; ++i) x[i] = 0.f;
        }
    } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(row + 16 * hh);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 v = src[c4];
            x[4 * c4] = v[0]; x[4 * c4 + 1] = v[1]; x[4 * c4 + 2] = v[2]; x[4 * c4 + 3] = v[3];
        }
    }
}

// acc = sum over tables.  OUT16 = false: (N,H,32) f32 rows, P16 input rows are widened (denominator moves to
// column D).  OUT16 = true (P16 input only): the sum is stored again as a packed 64-B row (numerators rounded to
// bf16 once more, denominator exact f32) -- the form table sharding exchanges between GPUs.
template <bool P16, bool OUT16>
__global__ __launch_bounds__(256) void reduce_tables_kernel(const float* __restrict__ part, int Tl, int D,
                                                            size_t n_rows, float* __restrict__ acc) {
    constexpr int ROWF = P16 ? 16 : 32;
    const size_t total = n_rows * 2;  // one lane per half row
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i >> 1;
        const int hh = (int)(i & 1);
        float s[16];
        load_half_row<P16>(part + row * ROWF, hh, s);
        for (int t = 1; t < Tl; ++t) {
            float x[16];
            load_half_row<P16>(part + row * ROWF + (size_t)t * n_rows * ROWF, hh, x);
#pragma unroll
            for (int u = 0; u < 16; ++u) s[u] += x[u];
        }
        if constexpr (OUT16) {
            u32x4* dst = reinterpret_cast<u32x4*>(acc + row * 16 + 8 * hh);
            if (hh == 0) {
                dst[0] = u32x4{hept_pack_bf16(s[0], s[1]), hept_pack_bf16(s[2], s[3]), hept_pack_bf16(s[4], s[5]),
                               hept_pack_bf16(s[6], s[7])};
                dst[1] = u32x4{hept_pack_bf16(s[8], s[9]), hept_pack_bf16(s[10], s[11]), hept_pack_bf16(s[12], s[13]),
                               hept_pack_bf16(s[14], s[15])};
            } else {  // widened columns 16..23 = numerators, 24 = denominator
                dst[0] = u32x4{hept_pack_bf16(s[0], s[1]), hept_pack_bf16(s[2], s[3]), hept_pack_bf16(s[4], s[5]),
                               hept_pack_bf16(s[6], s[7])};
                dst[1] = u32x4{__float_as_uint(s[8]), 0u, 0u, 0u};
            }
        } else {
            f32x4* dst = reinterpret_cast<f32x4*>(acc + row * 32 + 16 * hh);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) dst[c4] = f32x4{s[4 * c4], s[4 * c4 + 1], s[4 * c4 + 2], s[4 * c4 + 3]};
        }
    }
}

// Table sharding, one head group at a time: dst (n_pad, hg, row) = sum over tables of the rows of heads [h0, h0 + hg)
// of part (Tl, N, H, row); points >= N (padding up to a multiple of the rank count) get zero rows.  Same arithmetic
// and row formats as reduce_tables_kernel.
template <bool P16, bool OUT16>
__global__ __launch_bounds__(256) void reduce_heads_kernel(const float* __restrict__ part, int Tl, int N, int H, int h0,
                                                           int hg, int n_pad, float* __restrict__ dst) {
    constexpr int ROWF = P16 ? 16 : 32;
    constexpr int OUTF = OUT16 ? 16 : 32;
    const size_t total = (size_t)n_pad * hg * 2;  // one lane per half row
    const size_t tstride = (size_t)N * H * ROWF;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t orow = i >> 1;
        const int hh = (int)(i & 1);
        const size_t n = orow / hg;
        const int hl = (int)(orow - n * hg);
        float s[16];
        if (n < (size_t)N) {
            const float* src = part + (n * H + h0 + hl) * ROWF;
            load_half_row<P16>(src, hh, s);
            for (int t = 1; t < Tl; ++t) {
                float x[16];
                load_half_row<P16>(src + (size_t)t * tstride, hh, x);
#pragma unroll
                for (int u = 0; u < 16; ++u) s[u] += x[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) s[u] = 0.f;
        }
        if constexpr (OUT16) {
            u32x4* o = reinterpret_cast<u32x4*>(dst + orow * OUTF + 8 * hh);
            o[0] = u32x4{hept_pack_bf16(s[0], s[1]), hept_pack_bf16(s[2], s[3]), hept_pack_bf16(s[4], s[5]),
                         hept_pack_bf16(s[6], s[7])};
            if (hh == 0)
                o[1] = u32x4{hept_pack_bf16(s[8], s[9]), hept_pack_bf16(s[10], s[11]), hept_pack_bf16(s[12], s[13]),
                             hept_pack_bf16(s[14], s[15])};
            else  // widened columns 16..23 = numerators, 24 = denominator
                o[1] = u32x4{__float_as_uint(s[8]), 0u, 0u, 0u};
        } else {
            f32x4* o = reinterpret_cast<f32x4*>(dst + orow * OUTF + 16 * hh);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) o[c4] = f32x4{s[4 * c4], s[4 * c4 + 1], s[4 * c4 + 2], s[4 * c4 + 3]};
        }
    }
}

// ---- backward of combine_out (training path): out = bias + W . (numer / den) -----------------------
// Two kernels.  Rows: lane (point, head) recomputes per_head = numer / den, pulls g_out back through its head's
// D x D weight slab (LDS, packed fp32 FMAs) and writes the gradient of the partial row [d numer | d den | 0].
// Weight: one lane per output column (h, j) walks a slice of points and keeps the D partial sums of
// dW[:, h*D + j] in registers; every workgroup writes its sums to scratch and a second kernel adds the workgroups
// in index order (float atomics made dW, db depend on the order in which workgroups retire).
constexpr int CB_D = 24, CB_PITCH = CB_D * CB_D + 4;  // head pitch = 4 (mod 32): conflict-free 16-B reads for 8 heads

__global__ __launch_bounds__(256) void combine_bwd_rows_kernel(const float* __restrict__ acc,
                                                               const float* __restrict__ g_out,
                                                               const float* __restrict__ W, int N, int H,
                                                               float* __restrict__ gacc) {
    __shared__ __attribute__((aligned(16))) float wt_s[8 * CB_PITCH];  // [h][c][j]
    const int tid = threadIdx.x, HD = H * CB_D;
    for (int i = tid; i < CB_D * HD; i += 256) {  // W (D, H*D) row-major: coalesced read
        const int c = i / HD, h = (i % HD) / CB_D, j = i % CB_D;
        wt_s[h * CB_PITCH + c * CB_D + j] = W[i];
    }
    __syncthreads();
    const size_t n_rows = (size_t)N * H;
    for (size_t row = (size_t)blockIdx.x * 256 + tid; row < n_rows; row += (size_t)gridDim.x * 256) {
        const int h = (int)(row % H);
        const size_t n = row / H;
        float g[CB_D], a[28];
        const f32x4* gs = reinterpret_cast<const f32x4*>(g_out + n * CB_D);
        const f32x4* as = reinterpret_cast<const f32x4*>(acc + row * 32);
#pragma unroll
        for (int i = 0; i < CB_D / 4; ++i) {
            const f32x4 v = gs[i];
            g[4 * i] = v[0]; g[4 * i + 1] = v[1]; g[4 * i + 2] = v[2]; g[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const f32x4 v = as[i];
            a[4 * i] = v[0]; a[4 * i + 1] = v[1]; a[4 * i + 2] = v[2]; a[4 * i + 3] = v[3];
        }
        const float inv = 1.0f / a[CB_D];
        const float* wh = wt_s + h * CB_PITCH;
        f32x4 dph[CB_D / 4];
#pragma unroll
        for (int jg = 0; jg < CB_D / 4; ++jg) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CB_D; ++c) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wh + c * CB_D + 4 * jg);
                s4 = __builtin_elementwise_fma(w4, f32x4{g[c], g[c], g[c], g[c]}, s4);
            }
            dph[jg] = s4;
            __builtin_amdgcn_sched_barrier(0);  // keep the 144 weight reads from being hoisted (register pressure)
        }
        // d numer[j] = dph[j] / den;   d den = - sum_j dph[j] * numer[j] / den^2
        float dot = 0.f;
        f32x4* out = reinterpret_cast<f32x4*>(gacc + row * 32);
#pragma unroll
        for (int jg = 0; jg < CB_D / 4; ++jg) {
#pragma unroll
            for (int u = 0; u < 4; ++u) dot = fmaf(dph[jg][u], a[4 * jg + u], dot);
            out[jg] = f32x4{dph[jg][0] * inv, dph[jg][1] * inv, dph[jg][2] * inv, dph[jg][3] * inv};
        }
        out[6] = f32x4{-dot * inv * inv, 0.f, 0.f, 0.f};
        out[7] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

constexpr int CBW_POINTS = 128;  // points per workgroup of the weight-gradient kernel
constexpr int CBW_GROUPS = 4;    // thread groups of H*D = 192 columns that share a workgroup's points
// d_weight[c][h*D+j] = sum_n g_out[n][c] * numer[n][h][j] / denom[n][h]: thread = column (h, j) of one group, the
// groups take every CBW_GROUPS-th point (4 points in flight per thread: the loop is bound by load latency, not by
// its 24 fma per point), fold through LDS, and one group adds the workgroup's sums to the output with atomics.
__global__ __launch_bounds__(192 * CBW_GROUPS) void combine_bwd_weight_kernel(const float* __restrict__ acc,
                                                                              const float* __restrict__ g_out, int N,
                                                                              int H, float* __restrict__ partial) {
    // one LDS region, used twice: the workgroup's g_out rows during the loop (every lane of a wave reads the same
    // word: an LDS broadcast instead of 24 scalar loads per point), then the fold of the thread groups
    __shared__ float red_s[CBW_GROUPS - 1][CB_D + 1][192];
    static_assert(sizeof(float) * CBW_POINTS * CB_D <= sizeof(float) * (CBW_GROUPS - 1) * (CB_D + 1) * 192, "g tile fits");
    float* g_s = &red_s[0][0][0];
    const int col = threadIdx.x % 192, grp = threadIdx.x / 192;  // col = h * D + j; 192 = 3 waves, so grp is wave-uniform
    const int HD = H * CB_D;
    const bool live = col < HD;
    const int h = live ? col / CB_D : 0, j = col % CB_D;
    const int n_begin = blockIdx.x * CBW_POINTS, n_end = min(N, n_begin + CBW_POINTS);
    float s[CB_D];
#pragma unroll
    for (int c = 0; c < CB_D; ++c) s[c] = 0.f;
    float sb = 0.f;  // lanes < D also sum g_out[:, lane] for the bias gradient
    for (int i = threadIdx.x; i < (n_end - n_begin) * CB_D; i += 192 * CBW_GROUPS) g_s[i] = g_out[(size_t)n_begin * CB_D + i];
    __syncthreads();
    constexpr int UN = 4;
    for (int n0 = n_begin + grp; n0 < n_end; n0 += CBW_GROUPS * UN) {
        float ph[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + u * CBW_GROUPS;
            const float* arow = acc + ((size_t)(n < n_end ? n : n_begin) * H + h) * 32;
            ph[u] = n < n_end ? arow[j] / arow[CB_D] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + u * CBW_GROUPS;
            if (n < n_end) {                                 // wave-uniform
                const float* gr = g_s + (n - n_begin) * CB_D;  // uniform across the wave: LDS broadcast
#pragma unroll
                for (int c = 0; c < CB_D; ++c) s[c] = fmaf(gr[c], ph[u], s[c]);
                if (col < CB_D) sb += gr[col];
            }
        }
    }
    __syncthreads();  // every group is done with the g tile
    if (grp > 0) {
#pragma unroll
        for (int c = 0; c < CB_D; ++c) red_s[grp - 1][c][col] = s[c];
        red_s[grp - 1][CB_D][col] = sb;
    }
    __syncthreads();
    if (grp == 0 && live) {
#pragma unroll
        for (int g2 = 0; g2 < CBW_GROUPS - 1; ++g2) {
#pragma unroll
            for (int c = 0; c < CB_D; ++c) s[c] += red_s[g2][c][col];
            sb += red_s[g2][CB_D][col];
        }
        float* mine = partial + (size_t)blockIdx.x * (CB_D + 1) * 192;   // [c][col], row CB_D = bias sums
#pragma unroll
        for (int c = 0; c < CB_D; ++c) mine[c * 192 + col] = s[c];
        mine[CB_D * 192 + col] = col < CB_D ? sb : 0.f;
    }
}

// d_weight[c][col] = sum over workgroups of partial[wg][c][col], d_bias[col] likewise from row CB_D: workgroups added
// in a fixed order (hept_fixed_sum)
__global__ __launch_bounds__(256) void combine_bwd_weight_sum_kernel(const float* __restrict__ partial, int n_wgs, int HD,
                                                                     float* __restrict__ d_weight,
                                                                     float* __restrict__ d_bias) {
    __shared__ float red_s[HEPT_FSUM_SLICES * HEPT_FSUM_OUT];
    constexpr int TOTAL = (CB_D + 1) * 192;
    const int i = blockIdx.x * HEPT_FSUM_OUT + threadIdx.x % HEPT_FSUM_OUT;
    const int c = i / 192, col = i - c * 192;
    const bool valid = i < TOTAL && col < HD;
    const float tot = hept_fixed_sum(partial, n_wgs, TOTAL, i, valid, red_s);
    if (threadIdx.x < HEPT_FSUM_OUT && valid) {
        if (c < CB_D) d_weight[(size_t)c * HD + col] = tot;
        else if (col < CB_D && d_bias) d_bias[col] = tot;
    }
}

// ---- the same backward for any head count / head dimension (H <= 16, D <= 27; reference example/hept.py:34-41 accepts
// any): the same formulas with run-time loops, one thread per (point, head) row / per column of d_weight.  Not tuned.
__global__ __launch_bounds__(256) void combine_bwd_rows_generic_kernel(const float* __restrict__ acc,
                                                                       const float* __restrict__ g_out,
                                                                       const float* __restrict__ W, int N, int H, int D,
                                                                       float* __restrict__ gacc) {
    extern __shared__ float wg_s[];   // W (D, H*D) as stored
    const int tid = threadIdx.x, HD = H * D;
    for (int i = tid; i < D * HD; i += 256) wg_s[i] = W[i];
    __syncthreads();
    const size_t n_rows = (size_t)N * H;
    for (size_t row = (size_t)blockIdx.x * 256 + tid; row < n_rows; row += (size_t)gridDim.x * 256) {
        const int h = (int)(row % H);
        const size_t n = row / H;
        const float* a = acc + row * 32;
        const float* g = g_out + n * D;
        float* out = gacc + row * 32;
        const float inv = 1.0f / a[D];
        float dot = 0.f;
        for (int j = 0; j < D; ++j) {
            float s = 0.f;
            for (int c = 0; c < D; ++c) s = fmaf(wg_s[c * HD + h * D + j], g[c], s);
            dot = fmaf(s, a[j], dot);
            out[j] = s * inv;
        }
        out[D] = -dot * inv * inv;
        for (int j = D + 1; j < 32; ++j) out[j] = 0.f;
    }
}

// partial[wg][c][col] (c == D: bias sums in columns < D), pitch (D + 1) * H * D per workgroup of CBW_POINTS points
__global__ __launch_bounds__(256) void combine_bwd_weight_generic_kernel(const float* __restrict__ acc,
                                                                         const float* __restrict__ g_out, int N, int H,
                                                                         int D, float* __restrict__ partial) {
    __shared__ float g_s[CBW_POINTS * 27];
    const int HD = H * D;
    const int n_begin = blockIdx.x * CBW_POINTS, n_end = min(N, n_begin + CBW_POINTS), cnt = n_end - n_begin;
    for (int i = threadIdx.x; i < cnt * D; i += 256) g_s[i] = g_out[(size_t)n_begin * D + i];
    __syncthreads();
    float* mine = partial + (size_t)blockIdx.x * (D + 1) * HD;
    for (int col = threadIdx.x; col < HD; col += 256) {
        const int h = col / D, j = col - h * D;
        float s[27];
#pragma unroll
        for (int c = 0; c < 27; ++c) s[c] = 0.f;
        float sb = 0.f;
        for (int n = n_begin; n < n_end; ++n) {
            const float* arow = acc + ((size_t)n * H + h) * 32;
            const float ph = arow[j] / arow[D];
            const float* gr = g_s + (n - n_begin) * D;
#pragma unroll
            for (int c = 0; c < 27; ++c)
                if (c < D) s[c] = fmaf(gr[c], ph, s[c]);
            if (col < D) sb += gr[col];
        }
#pragma unroll
        for (int c = 0; c < 27; ++c)
            if (c < D) mine[c * HD + col] = s[c];
        mine[D * HD + col] = col < D ? sb : 0.f;
    }
}

__global__ __launch_bounds__(256) void combine_bwd_weight_sum_generic_kernel(const float* __restrict__ partial, int n_wgs,
                                                                             int HD, int D, float* __restrict__ d_weight,
                                                                             float* __restrict__ d_bias) {
    __shared__ float red_s[HEPT_FSUM_SLICES * HEPT_FSUM_OUT];
    const int total = (D + 1) * HD;
    const int i = blockIdx.x * HEPT_FSUM_OUT + threadIdx.x % HEPT_FSUM_OUT;
    const bool valid = i < total;
    const float tot = hept_fixed_sum(partial, n_wgs, (size_t)total, i, valid, red_s);
    if (threadIdx.x < HEPT_FSUM_OUT && valid) {
        const int c = i / HD, col = i - c * HD;
        if (c < D) d_weight[(size_t)c * HD + col] = tot;
        else if (col < D && d_bias) d_bias[col] = tot;
    }
}

}  // namespace

extern "C" int hept_reduce_tables(const float* part, int part_precision, int Tl, int N, int H, int D, float* acc,
                                  int acc_precision, void* stream) {
    if (!part || !acc) return HEPT_ERR_ARG;
    if (acc_precision != HEPT_PREC_F32 && !(acc_precision == HEPT_PREC_BF16 && part_precision == HEPT_PREC_BF16))
        return HEPT_ERR_SHAPE;
    if (Tl < 1 || N < 1 || H < 1 || D < 1 || D > 28) return HEPT_ERR_SHAPE;
    const size_t n_rows = (size_t)N * H;
    const size_t blocks = (n_rows * 2 + 255) / 256;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (part_precision == HEPT_PREC_BF16) {
        if (D != 24) return HEPT_ERR_SHAPE;  // packed rows exist for D == 24 only
        if (acc_precision == HEPT_PREC_BF16)
            hipLaunchKernelGGL((reduce_tables_kernel<true, true>), dim3(grid), dim3(256), 0, st, part, Tl, D, n_rows, acc);
        else
            hipLaunchKernelGGL((reduce_tables_kernel<true, false>), dim3(grid), dim3(256), 0, st, part, Tl, D, n_rows, acc);
    } else if (part_precision == HEPT_PREC_F32) {
        hipLaunchKernelGGL((reduce_tables_kernel<false, false>), dim3(grid), dim3(256), 0, st, part, Tl, D, n_rows, acc);
    } else {
        return HEPT_ERR_SHAPE;
    }
    return hept_launch_status();
}

extern "C" int hept_reduce_heads(const float* part, int part_precision, int Tl, int N, int H, int D, int h0, int hg,
                                 int n_pad, float* dst, int acc_precision, void* stream) {
    if (!part || !dst) return HEPT_ERR_ARG;
    if (acc_precision != HEPT_PREC_F32 && !(acc_precision == HEPT_PREC_BF16 && part_precision == HEPT_PREC_BF16))
        return HEPT_ERR_SHAPE;
    if (Tl < 1 || N < 1 || H < 1 || D < 1 || D > 28 || h0 < 0 || hg < 1 || h0 + hg > H || n_pad < N)
        return HEPT_ERR_SHAPE;
    const size_t blocks = ((size_t)n_pad * hg * 2 + 255) / 256;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (part_precision == HEPT_PREC_BF16) {
        if (D != 24) return HEPT_ERR_SHAPE;
        if (acc_precision == HEPT_PREC_BF16)
            hipLaunchKernelGGL((reduce_heads_kernel<true, true>), dim3(grid), dim3(256), 0, st, part, Tl, N, H, h0, hg, n_pad, dst);
        else
            hipLaunchKernelGGL((reduce_heads_kernel<true, false>), dim3(grid), dim3(256), 0, st, part, Tl, N, H, h0, hg, n_pad, dst);
    } else if (part_precision == HEPT_PREC_F32) {
        hipLaunchKernelGGL((reduce_heads_kernel<false, false>), dim3(grid), dim3(256), 0, st, part, Tl, N, H, h0, hg, n_pad, dst);
    } else {
        return HEPT_ERR_SHAPE;
    }
    return hept_launch_status();
}

// few tiles: one tile per workgroup, head pairs split over its waves (SPLIT); else one tile per wave
#ifndef HEPT_CMB_SPLIT_BELOW
#define HEPT_CMB_SPLIT_BELOW 1024
#endif
#ifndef HEPT_CMB_SPLIT_GRID
#define HEPT_CMB_SPLIT_GRID (1 << 30)
#endif
constexpr int CMB_SPLIT_BELOW = HEPT_CMB_SPLIT_BELOW;  // tiles; 1024 tiles = one wave per SIMD on 256 CUs

// HEPT_NO_STAGED_COMBINE=1: packed rows are loaded lane by lane as in round 3 (A/B measurements; read once)
inline bool staged_combine_off() {
    static const bool off = [] { const char* e = getenv("HEPT_NO_STAGED_COMBINE"); return e && *e && *e != '0'; }();
    return off;
}

constexpr int CMB_NO_LDS = -1;   // internal: the device refused the staged rows' LDS size (never leaves this file)
template <bool P16, bool FFN, int DT, bool PUSH, bool STG>
int combine_launch_impl(hipStream_t st, const float* part, int Tl, int N, int H, int D, int n0, int n_count,
                        const float* W, const float* b, float* out, const FfnIn& ffn, int HG, size_t gstride,
                        const P2pDev& px) {
    const int n_tiles = (n_count + 31) / 32;
    const bool split = n_tiles < CMB_SPLIT_BELOW;
    constexpr bool WDIRECT = (P16 || STG) && DT == 24;   // as in the kernel: no weight slab in LDS
    size_t lds = sizeof(float) * ((WDIRECT ? (size_t)0 : (size_t)((H + 1) & ~1) * 28 * WT_PITCH) +
                                  (FFN ? FFN_WFLOATS + CMB_WAVES * 32 * FFN_PITCH : 0) +
                                  (split ? (CMB_WAVES - 1) * 16 * 64 : 0) +
                                  (PUSH ? (split ? 1 : CMB_WAVES) * 32 * 24 : 0));
    lds = (lds + 15) & ~(size_t)15;
    const int stg_off = (int)lds;
    if (STG) lds += (size_t)CMB_WAVES * (P16 ? STG_WAVE_BYTES : STG32_WAVE_BYTES);
    if (lds > 65536) {   // many heads (the weight slab alone is 3.6 KiB per head), or the staged rows' images
        static LdsRaised raised_split, raised_flat;
        if (hept_raise_lds(split ? raised_split : raised_flat,
                           split ? reinterpret_cast<const void*>(&combine_out_kernel<P16, FFN, DT, true, PUSH, STG>)
                                 : reinterpret_cast<const void*>(&combine_out_kernel<P16, FFN, DT, false, PUSH, STG>), lds))
            return STG ? CMB_NO_LDS : HEPT_ERR_LAUNCH;   // (staged rows: the caller takes the lane-by-lane kernel instead)
    }
    if (split) {
        hipLaunchKernelGGL((combine_out_kernel<P16, FFN, DT, true, PUSH, STG>), dim3(n_tiles < HEPT_CMB_SPLIT_GRID ? n_tiles : HEPT_CMB_SPLIT_GRID), dim3(CMB_THREADS), lds, st, part,
                           Tl, N, H, D, n0, n_count, W, b, out, HG, gstride, ffn, px, stg_off);
    } else {
        const int wgs = (n_tiles + CMB_WAVES - 1) / CMB_WAVES;
#ifndef HEPT_CMB_MAX_WGS
#define HEPT_CMB_MAX_WGS 2048
#endif
        hipLaunchKernelGGL((combine_out_kernel<P16, FFN, DT, false, PUSH, STG>), dim3(wgs < HEPT_CMB_MAX_WGS ? wgs : HEPT_CMB_MAX_WGS), dim3(CMB_THREADS),
                           lds, st, part, Tl, N, H, D, n0, n_count, W, b, out, HG, gstride, ffn, px, stg_off);
    }
    return hept_launch_status();
}

template <bool P16, bool FFN, int DT, bool PUSH = false>
int combine_launch(hipStream_t st, const float* part, int Tl, int N, int H, int D, int n0, int n_count, const float* W,
                   const float* b, float* out, const FfnIn& ffn, int HG = 0, size_t gstride = 0,
                   const P2pDev& px = P2pDev{}) {
    if (HG <= 0) HG = H;
    // D = 24 rows are read as 16-B pieces (direct-to-LDS loads of whole rows, the weight column as f32x4 loads): a base
    // that is not 16-B aligned would fault inside the kernel instead of failing here (include/hept_hip.h says so)
    if (DT == 24 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(part)) & 15)) return HEPT_ERR_ARG;
    // staged rows: an even number of heads per head group (a staged unit is the two rows of a head pair, contiguous in
    // a point's slot) -- the plain layout, and the receive region of the table-sharded step (PUSH; "tables" = source ranks)
    if constexpr (DT == 24) {
        if (HG % 2 == 0 && H % 2 == 0 && !staged_combine_off()) {
            const int rc = combine_launch_impl<P16, FFN, DT, PUSH, true>(st, part, Tl, N, H, D, n0, n_count, W, b, out, ffn, HG, gstride, px);
            if (rc != CMB_NO_LDS) return rc;   // (the raise of the dynamic LDS limit failed: lane-by-lane loads need less)
        }
    }
    return combine_launch_impl<P16, FFN, DT, PUSH, false>(st, part, Tl, N, H, D, n0, n_count, W, b, out, ffn, HG, gstride, px);
}

extern "C" int hept_combine_groups(const float* part, int part_precision, int Tl, int N, int H, int D, int n0,
                                   int n_count, int HG, size_t group_stride, const float* out_weight,
                                   const float* out_bias, float* out, void* stream) {
    if (!part || !out_weight || !out) return HEPT_ERR_ARG;
    if (Tl < 1 || N < 1 || H < 1 || H > 16 || D < 1 || D > 27 || n0 < 0 || n_count < 0 || n0 + n_count > N)
        return HEPT_ERR_SHAPE;
    if (HG < 1 || HG > H || H % HG != 0) return HEPT_ERR_SHAPE;
    if (n_count == 0) return HEPT_OK;
    hipStream_t st = (hipStream_t)stream;
    const FfnIn none{};
    const size_t gs = group_stride;
    if (part_precision == HEPT_PREC_BF16) {
        if (D != 24) return HEPT_ERR_SHAPE;  // packed rows keep the denominator at widened column 24
        return combine_launch<true, false, 24>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, out, none, HG, gs);
    }
    if (part_precision != HEPT_PREC_F32) return HEPT_ERR_SHAPE;
    if (D == 24) return combine_launch<false, false, 24>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, out, none, HG, gs);
    if (D == 16) return combine_launch<false, false, 16>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, out, none, HG, gs);
    return combine_launch<false, false, 0>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, out, none, HG, gs);
}

// one-sided table sharding: wait for the received rows, combine this rank's points, store them into every rank's
// gathered output and raise the output flag (internal: called by hept_forward_sharded; D == 24, n_count >= 1)
int hept_combine_push(const float* part, int part_precision, int Tl, int N, int H, int n_count, int HG,
                      size_t group_stride, const float* out_weight, const float* out_bias, const P2pDev& px,
                      hipStream_t st) {
    const FfnIn none{};
    if (part_precision == HEPT_PREC_BF16)
        return combine_launch<true, false, 24, true>(st, part, Tl, N, H, 24, 0, n_count, out_weight, out_bias, nullptr, none,
                                                     HG, group_stride, px);
    if (part_precision == HEPT_PREC_F32)
        return combine_launch<false, false, 24, true>(st, part, Tl, N, H, 24, 0, n_count, out_weight, out_bias, nullptr, none,
                                                      HG, group_stride, px);
    return HEPT_ERR_SHAPE;
}

extern "C" int hept_combine_out(const float* part, int part_precision, int Tl, int N, int H, int D, int n0,
                                int n_count, const float* out_weight, const float* out_bias, float* out,
                                void* stream) {
    return hept_combine_groups(part, part_precision, Tl, N, H, D, n0, n_count, H, 0, out_weight, out_bias, out, stream);
}

extern "C" int hept_combine_ffn(const float* part, int part_precision, int Tl, int N, int H, int D, int n0,
                                int n_count, const float* out_weight, const float* out_bias, const float* x,
                                const float* norm_w, const float* norm_b, float eps, const float* ff1_w,
                                const float* ff1_b, const float* ff2_w, const float* ff2_b, float* y, void* stream) {
    if (!part || !out_weight || !x || !norm_w || !norm_b || !ff1_w || !ff1_b || !ff2_w || !ff2_b || !y)
        return HEPT_ERR_ARG;
    if (Tl < 1 || N < 1 || H < 1 || H > 16 || D != FFN_D || n0 < 0 || n_count < 0 || n0 + n_count > N)
        return HEPT_ERR_SHAPE;
    if (n_count == 0) return HEPT_OK;
    hipStream_t st = (hipStream_t)stream;
    const FfnIn ffn{x, norm_w, norm_b, ff1_w, ff1_b, ff2_w, ff2_b, eps};
    if (part_precision == HEPT_PREC_BF16)
        return combine_launch<true, true, 24>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, y, ffn);
    if (part_precision == HEPT_PREC_F32)
        return combine_launch<false, true, 24>(st, part, Tl, N, H, D, n0, n_count, out_weight, out_bias, y, ffn);
    return HEPT_ERR_SHAPE;
}

static bool combine_bwd_tuned(int H, int D) { return D == CB_D && H <= 8; }   // the shipped models' rows

extern "C" size_t hept_combine_bwd_scratch_bytes_shape(int N, int H, int D) {
    if (N < 1 || H < 1 || D < 1) return 0;
    const size_t wgs = (size_t)((N + CBW_POINTS - 1) / CBW_POINTS);
    return wgs * (combine_bwd_tuned(H, D) ? (size_t)(CB_D + 1) * 192 : (size_t)(D + 1) * H * D) * sizeof(float);
}

extern "C" size_t hept_combine_bwd_scratch_bytes(int N) { return hept_combine_bwd_scratch_bytes_shape(N, 8, CB_D); }

extern "C" int hept_combine_bwd(const float* acc, const float* g_out, const float* out_weight, int N, int H, int D,
                                float* gacc, float* d_weight, float* d_bias, void* scratch, size_t scratch_bytes,
                                void* stream) {
    if (!acc || !g_out || !out_weight || !gacc || !d_weight || !scratch) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || H > 16 || D < 1 || D > 27) return HEPT_ERR_SHAPE;
    if (scratch_bytes < hept_combine_bwd_scratch_bytes_shape(N, H, D)) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n_rows = (size_t)N * H;
    const size_t blocks = (n_rows + 255) / 256;
    const int n_wgs = (N + CBW_POINTS - 1) / CBW_POINTS;
    float* partial = static_cast<float*>(scratch);
    if (!combine_bwd_tuned(H, D)) {
        const size_t lds = (size_t)D * H * D * sizeof(float);   // <= 46.7 KB
        hipLaunchKernelGGL(combine_bwd_rows_generic_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), lds,
                           st, acc, g_out, out_weight, N, H, D, gacc);
        hipLaunchKernelGGL(combine_bwd_weight_generic_kernel, dim3(n_wgs), dim3(256), 0, st, acc, g_out, N, H, D, partial);
        const int total = (D + 1) * H * D;
        hipLaunchKernelGGL(combine_bwd_weight_sum_generic_kernel, dim3((total + HEPT_FSUM_OUT - 1) / HEPT_FSUM_OUT),
                           dim3(256), 0, st, partial, n_wgs, H * D, D, d_weight, d_bias);
        return hept_launch_status();
    }
    hipLaunchKernelGGL(combine_bwd_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, acc,
                       g_out, out_weight, N, H, gacc);
    hipLaunchKernelGGL(combine_bwd_weight_kernel, dim3(n_wgs), dim3(192 * CBW_GROUPS), 0, st, acc, g_out, N, H, partial);
    hipLaunchKernelGGL(combine_bwd_weight_sum_kernel, dim3((CB_D + 1) * 192 / HEPT_FSUM_OUT), dim3(256), 0, st, partial,
                       n_wgs, H * D, d_weight, d_bias);
    return hept_launch_status();
}
