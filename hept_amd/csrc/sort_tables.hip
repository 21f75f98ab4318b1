// sort_tables: AND-shifted sort keys + exact stable segmented sort: one counting pass on the top bits of a monotone
// bucket id, then every bucket is finished inside LDS.  Two kernels; the second one also carries the v rows of the row
// builder as rider workgroups (RowsJob below).
//
// Replaces, for tables [t0, t0+Tl) (reference file:line):
//   hash_shift = max - min            example/hept_utils.py:70
//   key = hash + float(code) * shift  example/hept.py:63-65   (two rounded ops, no FMA)
//   argsort(dim=-1) x 2               example/hept.py:67-68
//
// 2*Tl*H independent segments of N fp32 keys (q segments first, then k).  The result is
// exactly torch.sort(stable=True): ascending key, ties in ascending point index (the
// reference's own argsort is unstable and leaves tie order undefined, SURVEY.md §7 hard part 1).
//
// Integer/byte work, HBM/L2-bound.  Every key gets a MONOTONE bucket id of HEPT_ID_BITS bits
//     id(key) = trunc((key - kmin) * 2^bits / (kmax - kmin)),
// where [kmin,kmax] = [hash min, hash max + largest code * span] is known before any key exists (from the
// prep kernel's partials).  id() is a monotone non-decreasing function of the key itself (fp32 subtract,
// multiply and truncate are monotone), so id order never contradicts key order and equal keys share an id.
//   KA chunk sort   one workgroup per 4096-key chunk of a segment: keys (two rounded ops) -> (key, low id, index) pairs,
//                   digit-sorted by the top HEPT_TOP_BITS id bits inside LDS (slot = LDS atomic on the digit counter)
//                   and written back IN PLACE as one linear run, next to the chunk's 257-entry digit offset table.
//                   No key array, no global histogram, no cross-workgroup prefix.
//   KB bucket sort  one workgroup per (segment, bucket): collects the bucket's pairs from the n_chunks runs (its final
//                   position range starts at the sum of the chunks' offsets of its digit), then histogram / prefix /
//                   scatter on the LOW id bits inside LDS groups the pairs by their full id, and every pair counts the
//                   members of its id group that are smaller as u64 (key << 32 | index) -> final position.  Groups are
//                   1-3 pairs.
// The result is the exact stable sort for ANY input; cost O(N + sum group^2).  Adversarial inputs (all keys
// inside 2^-bits of the range) degrade to O(N^2) compares per segment -- slow, never wrong.
// History (tracking-60k, 48 segments x 60 032 keys): 4-pass LSD radix 144 us -> two 8-bit LSD passes + windowed
// neighbour rank 89 us -> keygen / scatter / bucket sort 40 us (three launches, 11.5 MB of keys and 23 MB of pairs
// written and read back) -> this design (two launches, keys never leave the chip).
#include "common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / HEPT_WAVE;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int RADIX = 256;
#ifndef HEPT_ID_BITS
#define HEPT_ID_BITS 17
#endif
#ifndef HEPT_TOP_BITS
#define HEPT_TOP_BITS 8
#endif
#ifndef HEPT_BKT_THREADS
#define HEPT_BKT_THREADS 128
#endif
#ifndef HEPT_BKT_CAP
#define HEPT_BKT_CAP 1024
#endif
constexpr int ID_BUCKETS = 1 << HEPT_ID_BITS;       // resolution of the monotone bucket id (exact in fp32: <= 2^24)
constexpr int TOP_SHIFT = HEPT_ID_BITS - HEPT_TOP_BITS;  // K3 partitions by the top id bits ...
constexpr int NTOP = 1 << HEPT_TOP_BITS;            // ... into NTOP buckets per segment (digits < RADIX)
constexpr int LOBINS = 1 << TOP_SHIFT;              // K4 groups a bucket by the remaining low id bits
static_assert(NTOP <= RADIX, "the histogram / scan / scatter arrays hold RADIX digits");

__device__ __forceinline__ unsigned int ordered_bits(float key) {
    if (key == 0.f) key = 0.f;  // -0.0 and +0.0 compare equal in the reference sort
    const unsigned int u = __float_as_uint(key);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned int u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// monotone HEPT_ID_BITS-bit bucket id; scale = 2^bits / (kmax - kmin), 0 when all keys are equal
__device__ __forceinline__ unsigned int bucket_id(unsigned int u, float kmin, float scale) {
    // the float clamp keeps +inf keys (the src variant's padding rows) and inf * 0 = NaN in the last bucket
    const float x = fminf((from_ordered(u) - kmin) * scale, (float)(ID_BUCKETS - 1));
    const int b = (int)x;
    return (unsigned int)(b < 0 ? 0 : b);
}

// per-segment id map parameters, written once by KA (chunk 0) and read by KB
struct SegParams {
    float kmin, scale;
};

// REGION layout (segments of SMALL_CAP < N <= REGION_MAX_N keys; round 5): every (segment, bucket) owns a fixed region
// of `capb` pair slots, region[seg][bucket][0 .. capb), at a pitch that is an odd number of 256-B units (the regions'
// first lines must not all fall on the same memory channels).  KA reserves the slots of a chunk's run with ONE atomic
// per (chunk, bucket) on cnt[seg][bucket] -- the return value is where the run starts inside the region -- so KB reads
// one contiguous array instead of n_chunks runs behind a dependent table lookup, and the first final position of a
// bucket is the sum of the counts in front of it.  Arrival order inside a region is arbitrary: KB sorts unique
// (key, index) pairs, the result does not depend on it.  Pairs that do not fit (skewed keys, ties) go to the segment's
// overflow pool through a bump cursor; a bucket with more than capb pairs collects them from there (slow, exact).
// The counters (cnt[segs][NTOP], then cursor[segs]) must be zero when KA starts: the row builder of the same forward
// zeroes them (hept_sort_zero_block), stand-alone callers pay a fill.
struct RegionArgs {
    unsigned int* cnt;               // [segs][NTOP] pairs per bucket, then [segs] overflow-pool cursors
    unsigned long long* region;      // [segs][NTOP][pitch]
    unsigned long long* pool;        // [segs][N] overflow pairs of the segment, any order
    unsigned long long* gathered;    // [segs][N] KB: the pairs of an overflowing bucket, collected (at the bucket's positions)
    unsigned int capb, pitch;        // slots per region; pitch in pairs
    int segs;
};
constexpr int REGION_MAX_N = 131072;   // = NTOP * BKT_CAP_SMALL / 2: the range of the small-tile bucket kernel
constexpr unsigned int region_cap(int N) {   // >= 8x the average bucket, a power of two, >= 1024 = 2 x the lean path's bucket
    unsigned int c = 1024;   // (round 5's floor of 2048 gave an 8k-point cloud 214 MB of regions: ADVICE round 5)
    while ((size_t)c * 32 < (size_t)N) c <<= 1;
    return c;
}
constexpr unsigned int region_pitch(unsigned int capb) { return capb + 32; }   // + 256 B: an odd multiple of 256 B

// EMBED (segments of at most 2^20 keys): the pair carries the low id bits next to the point index,
//     pair = key << 32 | low id << 20 | index,
// so that KB reads a pair's group with a shift instead of recomputing the float id map (KB is bound by VALU issue:
// three id evaluations per pair were 40 % of its instructions).  The order of the pairs as u64 is unchanged: the id is
// a monotone function of the key, equal keys carry equal ids.
constexpr int EMBED_SHIFT = 20;
constexpr unsigned int EMBED_INDEX_MASK = (1u << EMBED_SHIFT) - 1u;
static_assert(TOP_SHIFT + EMBED_SHIFT <= 32, "low id bits + index fit the low word of a pair");

// KA: one workgroup = one 4096-key chunk of one segment.  The launch is a single round of workgroups (720 at
// tracking-60k on 1024 resident slots), so its length is one workgroup's dependent chain: 512 threads halve the
// per-thread item loops of that chain and double the waves that hide its latencies.
//   loads of the chunk's hashes and codes (issued first: they do not depend on the key range)
//   -> key range of the segment (the prep kernel's per-workgroup partials, 16 KiB from L2)
//   -> keys: key = proj + float(code) * span as TWO rounded ops (the two eager ops of the reference,
//      example/hept.py:63-65; HIP's __fmul_rn / __fadd_rn are plain * and + and would be contracted to one fma
//      without -ffp-contract=off (Makefile) -- the asm barrier makes it explicit here as well)
//   -> digit = top id bits; local slot = LDS atomic on the digit counter; exclusive prefix over the digits
//   -> pairs into the LDS stage in digit order -> one linear, fully coalesced run of pairs[seg][chunk * 4096 ...]
//      + tab[seg][chunk][0..256] = first position of every digit inside the run (tab[..][256] = pairs of the chunk)
// MODE 0: example keys;  MODE 1: the src variant's float shift (get_geo_shift, src/models/attention/hept.py:46-56):
// shift = (phi * span) * cfac + eta * span, every operation rounded on its own, key bound from the third partial column
// (src_bound_kernel);  MODE 2: raw keys (hept_segmented_argsort), range from raw_range_kernel.
#ifndef HEPT_SCATTER_THREADS
#define HEPT_SCATTER_THREADS 512
#endif
constexpr int SCT = HEPT_SCATTER_THREADS;
constexpr int SCT_ITEMS = SORT_CHUNK / SCT;
constexpr int SCT_WAVES = SCT / HEPT_WAVE;
constexpr int TAB = RADIX + 1;   // digit offset table of a chunk
static_assert(SCT >= RADIX && SCT % RADIX == 0 && SORT_CHUNK % SCT == 0 && SCT_ITEMS % 4 == 0, "digit ownership / items per thread");
static_assert(HEPT_PREP_GRID % SCT == 0, "range partials per thread");
template <int MODE, bool EMBED, bool REGION>
__global__ __launch_bounds__(SCT) void chunk_sort_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ eta_idx, const float* __restrict__ phi_idx, const float* __restrict__ cfac,
    const float* __restrict__ minmax, const unsigned int* __restrict__ range_bits, float range_lo, float range_hi, int N,
    int H, int t0, int Tl, int n_chunks, const int* __restrict__ seg_len, SegParams* __restrict__ seg_params,
    unsigned long long* __restrict__ pairs, unsigned int* __restrict__ tab, RegionArgs rg) {
    __shared__ unsigned long long stage_s[SORT_CHUNK];   // the chunk, digit-sorted (32 KiB)
    __shared__ unsigned int cnt_s[RADIX];                // keys of the chunk per digit
    __shared__ unsigned int start_s[TAB];                // first local position of a digit
    __shared__ unsigned int wsum_s[RADIX / HEPT_WAVE];
    __shared__ float red_s[3][SCT_WAVES];
    // REGION: digit of every staged position; slot of local position lp inside its bucket's region = lp + delta[digit]
    __shared__ unsigned char dig_s[REGION ? SORT_CHUNK : 4];
    __shared__ int delta_s[REGION ? RADIX : 1];
    __shared__ unsigned int over_s[REGION ? RADIX + 2 : 1];   // overflow: pairs of a digit beyond the region; [RADIX]: any, [RADIX+1]: pool base
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool digit = tid < RADIX;                      // this thread owns digit `tid`
    const int seg = blockIdx.y, chunk = blockIdx.x;
    if (REGION && tid == 0) over_s[RADIX] = 0;
    const size_t seg_off = (size_t)seg * N;        // N = segment stride; len = keys that take part (ragged argsort)
    const int len = seg_len ? seg_len[seg] : N;
    const int base = chunk * SORT_CHUNK;
    if (digit) cnt_s[tid] = 0;

    // ---- the chunk's inputs: item (i, e) of this thread is key n = base + (i * SCT + tid) * 4 + e
    const int th = MODE == 2 ? 0 : seg % (Tl * H);   // local (table, head)
    const int t = th / H, h = th % H;
    const float* proj = MODE == 2 ? qproj + seg_off : ((seg >= Tl * H ? kproj : qproj) + (size_t)th * N);
    const size_t row_off = ((size_t)(t0 + t) * H + h) * N;
    float pj[SCT_ITEMS];
    long long cd[MODE == 0 ? SCT_ITEMS : 1];
    float ev[MODE == 1 ? SCT_ITEMS : 1], pv[MODE == 1 ? SCT_ITEMS : 1];
    const bool vec_ok = (N % 4) == 0;  // segment bases stay 16-B aligned
#pragma unroll
    for (int i = 0; i < SCT_ITEMS / 4; ++i) {
        const int n = base + (i * SCT + tid) * 4;
        if (vec_ok && n + 3 < len) {
            const f32x4 p4 = hept_ld<HEPT_NT_KA_IN>(reinterpret_cast<const f32x4*>(proj + n));
#pragma unroll
            for (int e = 0; e < 4; ++e) pj[4 * i + e] = p4[e];
            if constexpr (MODE == 0) {
                typedef __attribute__((ext_vector_type(2))) long long i64x2;
                const i64x2 c01 = hept_ld<HEPT_NT_CODES>(reinterpret_cast<const i64x2*>(codes + row_off + n));
                const i64x2 c23 = hept_ld<HEPT_NT_CODES>(reinterpret_cast<const i64x2*>(codes + row_off + n + 2));
                cd[4 * i] = c01[0]; cd[4 * i + 1] = c01[1]; cd[4 * i + 2] = c23[0]; cd[4 * i + 3] = c23[1];
            }
            if constexpr (MODE == 1) {
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(eta_idx + row_off + n);
                const f32x4 f4 = *reinterpret_cast<const f32x4*>(phi_idx + row_off + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { ev[4 * i + e] = e4[e]; pv[4 * i + e] = f4[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = n + e < len;
                pj[4 * i + e] = ok ? proj[n + e] : 0.f;
                if constexpr (MODE == 0) cd[4 * i + e] = ok ? codes[row_off + n + e] : 0;
                if constexpr (MODE == 1) {
                    ev[4 * i + e] = ok ? eta_idx[row_off + n + e] : 0.f;
                    pv[4 * i + e] = ok ? phi_idx[row_off + n + e] : 0.f;
                }
            }
        }
    }

    // ---- key range of the segment -> id map (every workgroup of a segment computes the same two numbers)
    float lo, span = 0.f, scale;
    if constexpr (MODE == 2) {
        float hi;
        if (range_bits) {
            const unsigned int lo_b = range_bits[2 * seg], nhi_b = range_bits[2 * seg + 1];
            const bool none = lo_b == 0xFFFFFFFFu && nhi_b == 0xFFFFFFFFu;  // no finite key in the segment
            lo = none ? 0.f : from_ordered(lo_b);
            hi = none ? 0.f : from_ordered(~nhi_b);
        } else {   // the caller knows bounds of its keys (any bounds give an exact sort; tight ones balanced buckets)
            lo = range_lo;
            hi = range_hi;
        }
        const float width = hi - lo;
        scale = width > 0.f ? (float)ID_BUCKETS / width : 0.f;
    } else {
        // hash range + largest code of this (table, head): reduce the prep kernel's per-workgroup partials
        float hi = -INFINITY, cmax = 0.f;
        lo = INFINITY;
        f32x4 m[HEPT_PREP_GRID / SCT];
#pragma unroll
        for (int i = 0; i < HEPT_PREP_GRID / SCT; ++i)
            m[i] = *reinterpret_cast<const f32x4*>(minmax + (((size_t)t * H + h) * HEPT_PREP_GRID + i * SCT + tid) * 4);
#pragma unroll
        for (int i = 0; i < HEPT_PREP_GRID / SCT; ++i) {
            lo = fminf(lo, m[i][0]);
            hi = fmaxf(hi, m[i][1]);
            cmax = fmaxf(cmax, m[i][2]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
            cmax = fmaxf(cmax, __shfl_xor(cmax, off));
        }
        if (lane == 0) { red_s[0][w] = lo; red_s[1][w] = hi; red_s[2][w] = cmax; }
        __syncthreads();
        lo = red_s[0][0]; hi = red_s[1][0]; cmax = red_s[2][0];
#pragma unroll
        for (int ww = 1; ww < SCT_WAVES; ++ww) {
            lo = fminf(lo, red_s[0][ww]); hi = fmaxf(hi, red_s[1][ww]); cmax = fmaxf(cmax, red_s[2][ww]);
        }
        span = hi - lo;
        // keys lie in [lo, hi + cmax*span] (codes >= 0); any key outside is clamped by bucket_id (still monotone)
        const float width = (hi + cmax * span) - lo;
        scale = width > 0.f ? (float)ID_BUCKETS / width : 0.f;
    }
    if (!(scale < 3.0e38f)) scale = 0.f;  // inf/nan guard for denormal widths: one bucket, still exact
    if (chunk == 0 && tid == 0) seg_params[seg] = SegParams{lo, scale};
    if (MODE == 2) __syncthreads();   // (the other modes passed a barrier in the range reduction: cnt_s is zero)

    // ---- keys, digits, local slots
    float cf = 0.f;
    if constexpr (MODE == 1) cf = cfac[(size_t)(t0 + t) * H + h];
    unsigned int key[SCT_ITEMS];
    unsigned short rank[SCT_ITEMS], lowid[SCT_ITEMS];
    unsigned char dig[SCT_ITEMS];
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int n = base + ((r >> 2) * SCT + tid) * 4 + (r & 3);
        float kf;
        if constexpr (MODE == 0) {
            float off = (float)cd[r] * span;
            asm volatile("" : "+v"(off));
            kf = pj[r] + off;
        } else if constexpr (MODE == 1) {
            float t1 = ev[r] * span;
            asm volatile("" : "+v"(t1));
            float t2 = pv[r] * span;
            asm volatile("" : "+v"(t2));
            t2 = t2 * cf;
            asm volatile("" : "+v"(t2));
            float t3 = t2 + t1;
            asm volatile("" : "+v"(t3));
            kf = pj[r] + t3;
        } else {
            kf = pj[r];
        }
        key[r] = ordered_bits(kf);
        const unsigned int id = bucket_id(key[r], lo, scale);
        const unsigned int dg = id >> TOP_SHIFT;
        lowid[r] = (unsigned short)(id & (unsigned int)(LOBINS - 1));
        dig[r] = (unsigned char)dg;
        rank[r] = n < len ? (unsigned short)atomicAdd(&cnt_s[dg], 1u) : (unsigned short)0;
    }
    __syncthreads();
    // digit `tid`: exclusive prefix over the digits -> first local position of the digit
    const unsigned int total = digit ? cnt_s[tid] : 0u;
    // REGION: reserve the run's slots in the bucket's region (the round trip of the atomic runs beside the prefix and the
    // staging below; its result is needed for the write-out only)
    unsigned int gbase = 0;
    if constexpr (REGION) {
        if (digit && total) gbase = atomicAdd(rg.cnt + (size_t)seg * NTOP + tid, total);
    }
    const unsigned int incl = hept_wave_scan_add(total);
    if (digit && lane == 63) wsum_s[w] = incl;
    __syncthreads();
    const int n_valid = max(0, min(SORT_CHUNK, len - base));
    unsigned int first = incl - total;
    if (digit) {
#pragma unroll
        for (int ww = 0; ww < RADIX / HEPT_WAVE; ++ww)
            if (ww < w) first += wsum_s[ww];
        start_s[tid] = first;
        if constexpr (!REGION) {
            unsigned int* my_tab = tab + ((size_t)seg * n_chunks + chunk) * TAB;
            my_tab[tid] = first;
            if (tid == RADIX - 1) my_tab[RADIX] = (unsigned int)n_valid;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int n = base + ((r >> 2) * SCT + tid) * 4 + (r & 3);
        if (n < len) {
            const unsigned int lp = start_s[dig[r]] + rank[r];
            stage_s[lp] = ((unsigned long long)key[r] << 32) |
                          (EMBED ? ((unsigned int)lowid[r] << EMBED_SHIFT) | (unsigned int)n : (unsigned int)n);
            if constexpr (REGION) dig_s[lp] = dig[r];
        }
    }
    if constexpr (REGION) {
        if (digit) {
            delta_s[tid] = (int)gbase - (int)first;
            const unsigned int end = gbase + total;
            const unsigned int ov = end > rg.capb ? end - (gbase > rg.capb ? gbase : rg.capb) : 0u;
            over_s[tid] = ov;
            if (ov) over_s[RADIX] = 1u;
        }
    }
    __syncthreads();
    if constexpr (REGION) {
        // write out: every run goes to its bucket's region (consecutive lanes: consecutive slots of a run)
        unsigned long long* reg = rg.region + (size_t)seg * NTOP * rg.pitch;
        if (over_s[RADIX] == 0u) {   // uniform: the common case, every run fits
#pragma unroll
            for (int r = 0; r < SCT_ITEMS; ++r) {
                const int lp = r * SCT + tid;
                if (lp < n_valid) {
                    const unsigned int d = dig_s[lp];
                    reg[(size_t)d * rg.pitch + (unsigned int)(lp + delta_s[d])] = stage_s[lp];
                }
            }
        } else {
            // some run crosses the end of its region: those pairs go to the segment's overflow pool, packed by an
            // exclusive prefix over the digits' overflow counts behind one bump of the pool cursor
            const unsigned int ov = digit ? over_s[tid] : 0u;
            const unsigned int oincl = hept_wave_scan_add(ov);
            if (digit && lane == 63) wsum_s[w] = oincl;
            __syncthreads();
            if (digit) {
                unsigned int ofirst = oincl - ov;
#pragma unroll
                for (int ww = 0; ww < RADIX / HEPT_WAVE; ++ww)
                    if (ww < w) ofirst += wsum_s[ww];
                over_s[tid] = ofirst;
                if (tid == RADIX - 1) over_s[RADIX + 1] = atomicAdd(rg.cnt + (size_t)rg.segs * NTOP + seg, ofirst + ov);
            }
            __syncthreads();
            unsigned long long* pool = rg.pool + seg_off + over_s[RADIX + 1];
#pragma unroll
            for (int r = 0; r < SCT_ITEMS; ++r) {
                const int lp = r * SCT + tid;
                if (lp < n_valid) {
                    const unsigned int d = dig_s[lp];
                    const unsigned int slot = (unsigned int)(lp + delta_s[d]);
                    if (slot < rg.capb) {
                        reg[(size_t)d * rg.pitch + slot] = stage_s[lp];
                    } else {
                        const unsigned int gb = (unsigned int)(delta_s[d] + (int)start_s[d]);   // the run's first slot
                        pool[over_s[d] + (slot - (gb > rg.capb ? gb : rg.capb))] = stage_s[lp];
                    }
                }
            }
        }
    } else {
        // write out: the chunk's run, linear
        unsigned long long* out = pairs + seg_off + base;
#pragma unroll
        for (int r = 0; r < SCT_ITEMS; ++r) {
            const int lp = r * SCT + tid;
            if (lp < n_valid) out[lp] = stage_s[lp];
        }
    }
}

// KB: every (segment, top-bits bucket) is finished by one workgroup.  A bucket holds the pairs whose id shares the
// top bits; id is monotone in the key, so the bucket's final positions are exactly its own range [start, start + nb)
// and only the order inside it is left.  Its pairs lie in n_chunks runs (one per chunk KA sorted):
//        run c = pairs[seg][c * 4096 + tab[c][bucket] .. c * 4096 + tab[c][bucket + 1]),     start = sum_c tab[c][bucket]
// -- no cross-workgroup prefix anywhere.  The runs are ~16 pairs each when a chunk's keys are spread over the whole key
// range (one cloud), but a batch of clouds puts a chunk's keys into ITS cloud's part of the range: a bucket then draws
// hundreds of pairs from two or three chunks.  So the pairs are dealt to the threads by their position in run order
// (position i -> thread i % threads: one binary search over the run offsets, kept in LDS, per position), whatever the
// run lengths are.  The bucket (up to CAP pairs) is loaded into registers with every load in flight at once; positions
// beyond CAP are re-read from L2 by the later passes.
// Then: histogram of the LOW id bits -> exclusive prefix -> scatter (any order): the pairs are grouped by their full
// id, group g = [first[g], first[g+1]), and
//        final position(i) = start + first[g] + #{ j in group g : pair_j < pair_i }        (pairs are unique u64)
// Groups are 1-3 pairs at tracking-60k.  Only the grouped copy lives in LDS; one bins array serves as histogram,
// exclusive prefix and scatter cursor: after the scatter cur[d] is one past the last slot of group d, i.e. group
// d = [cur[d-1], cur[d]).  A bucket of up to 2 CAP pairs (skewed key distributions -- pileup clouds of unequal size --
// put 1.5x to 2x the expected pairs into a few buckets) goes through the tile in two passes, lower half of the id bins
// first, if each half fits.  Anything else (a pile of equal or nearly equal keys, possibly next to a few spread ones,
// which neither those bits nor any linear map of the key range can separate) takes the same three steps through a
// global scratch copy, grouped by SPLITTERS -- LOBINS pairs sampled at regular positions of the bucket and sorted in
// LDS; a pair's bin is the number of splitters <= it (binary search).  Pairs are unique u64, so even 60 000 equal keys
// spread over the bins by their index (streaming; slower, never wrong); a tile-sized bucket of equal keys costs at most
// CAP^2 compares.
constexpr int BKT_THREADS = HEPT_BKT_THREADS;
constexpr int BKT_WAVES = BKT_THREADS / HEPT_WAVE;

// ---- riders: the v half of the kvhat rows, written by workgroups that travel in the KB launch -------------------------
// KB is a chain of short dependent steps per workgroup: it moves 35 MB in 21 us and leaves the memory system idle most
// of that time.  The v role of the row builder (prep_hash.hip: [v | 1.0 at column D | 0 ..] per (head, point)) is the
// opposite -- a pure format conversion of 77 MB with no dependence on anything the forward computes -- so the first
// `vy` rows of the KB grid do it instead of sorting: no LDS and no more registers than KB itself (the launch's
// resources stay KB's).  Same values as the v role (the same bf16 conversion).  Measured at tracking-60k (bf16 / f32
// tiles): row builder 55.2 -> 46.2 / 65.2 -> 53.2 us, sort 37.2 -> 44.3 / 38.0 -> 47.7 us, forward -2.3 us; the riders
// alone take ~20 us (a rider wave is its own chain of load -> convert -> store trips), so more of them (8 grid rows:
// +4 us) or fewer (2: +1 us) are both worse than 3, and the q / k roles that stay behind run at 3.9 TB/s instead of
// the three roles' 4.7 -- the v role was already filling their gaps.
struct RowsJob {
    const float* v;     // (N, H * D) fp32
    char* kvhat;        // (H, N, 2 * qrow bytes): v half at byte qrow of a row
    int N, raw_size;    // rows >= raw_size are padding: v = 0 (src variant)
    int H, D4;          // D / 4
    int f32;            // 1: f32 tile rows (qrow = 128 B), 0: 16-bit rows (qrow = 64 B)
    int vy;             // grid rows (of gridDim.x workgroups each) that are riders; 0: none
};
#ifndef HEPT_ROWS_RIDERS
#define HEPT_ROWS_RIDERS 3
#endif
template <int H_>   // 0: run-time head count
__device__ __forceinline__ void rows_rider(const RowsJob& jb, unsigned int wg, unsigned int n_wgs) {
    // A wave takes a tile of 8 consecutive points (one contiguous run of v) at a time.  A row half is PPR 16-B pieces
    // (4 of 8 bf16 values, or 8 of 4 floats); lane order inside the tile is [head][point][piece]: a store instruction
    // writes whole 64-B / 128-B row halves of consecutive points of one head -- the pattern of the row builder's
    // copy-out -- and piece k of row (h, n) reads v[n][h*D + 8k ..] (bf16) or [.. + 4k ..] (f32): elements below D are
    // data, element D is the 1.0 that makes P.V produce the denominator, the rest zeros.  All loads of a tile are in
    // flight before the first conversion (trips of 2 / 4 instructions: the riders must not raise the launch's register
    // count above KB's own); the 128-B lines of the run are shared by the wave's instructions through L1.
    const unsigned int H = H_ ? H_ : (unsigned int)jb.H, D = 4u * (unsigned int)jb.D4, HD = H * D;
    const unsigned int lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned int ntiles = ((unsigned int)jb.N + 7u) >> 3;
    const unsigned int wave = wg * BKT_WAVES + wv, n_waves = n_wgs * BKT_WAVES;
    if (jb.f32) {
        constexpr unsigned int PPR = 8, IT = 4;   // pieces per row half; wave instructions per trip (16 at H = 8 per tile: 4 trips)
        for (unsigned int tile = wave; tile < ntiles; tile += n_waves) {
            const unsigned int n0 = tile << 3;
            for (unsigned int i0 = 0; i0 < H * 8 * PPR; i0 += 64 * IT) {
                f32x4 x[IT];
                unsigned int hd[IT], n[IT], k[IT];
#pragma unroll
                for (unsigned int u = 0; u < IT; ++u) {
                    const unsigned int i = i0 + u * 64 + lane;
                    k[u] = i & (PPR - 1);
                    n[u] = n0 + ((i / PPR) & 7u);
                    hd[u] = i / (8 * PPR);
                    const bool data = hd[u] < H && n[u] < (unsigned int)jb.raw_size && 4 * k[u] < D;
                    x[u] = data ? hept_ld<HEPT_NT_RIDER_IN32>(reinterpret_cast<const f32x4*>(jb.v + (size_t)n[u] * HD + hd[u] * D + 4 * k[u]))
                                : f32x4{4 * k[u] == D ? 1.f : 0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (unsigned int u = 0; u < IT; ++u)
                    if (hd[u] < H && n[u] < (unsigned int)jb.N)
                        *reinterpret_cast<f32x4*>(jb.kvhat + ((size_t)hd[u] * jb.N + n[u]) * 256 + 128 + k[u] * 16) = x[u];
            }
        }
    } else {
        constexpr unsigned int PPR = 4, IT = 2;   // (register budget: the launch's VGPR count must stay KB's)
        for (unsigned int tile = wave; tile < ntiles; tile += n_waves) {
            const unsigned int n0 = tile << 3;
            for (unsigned int i0 = 0; i0 < H * 8 * PPR; i0 += 64 * IT) {
                f32x4 x[IT][2];
                unsigned int hd[IT], n[IT], k[IT];
#pragma unroll
                for (unsigned int u = 0; u < IT; ++u) {
                    const unsigned int i = i0 + u * 64 + lane;
                    k[u] = i & (PPR - 1);
                    n[u] = n0 + ((i / PPR) & 7u);
                    hd[u] = i / (8 * PPR);
                    const bool row = hd[u] < H && n[u] < (unsigned int)jb.raw_size;
                    const float* src = jb.v + (size_t)n[u] * HD + hd[u] * D + 8 * k[u];
                    x[u][0] = (row && 8 * k[u] < D) ? hept_ld<HEPT_NT_RIDER_IN16>(reinterpret_cast<const f32x4*>(src)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    x[u][1] = (row && 8 * k[u] + 4 < D) ? hept_ld<HEPT_NT_RIDER_IN16>(reinterpret_cast<const f32x4*>(src + 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    // the 1.0 of column D (D % 4 == 0: the first element of one of the two quads)
                    if (8 * k[u] == D) x[u][0][0] = 1.f;
                    if (8 * k[u] + 4 == D) x[u][1][0] = 1.f;
                }
#pragma unroll
                for (unsigned int u = 0; u < IT; ++u)
                    if (hd[u] < H && n[u] < (unsigned int)jb.N)
                        *reinterpret_cast<u32x4*>(jb.kvhat + ((size_t)hd[u] * jb.N + n[u]) * 128 + 64 + k[u] * 16) =
                            u32x4{hept_pack_bf16(x[u][0][0], x[u][0][1]), hept_pack_bf16(x[u][0][2], x[u][0][3]),
                                  hept_pack_bf16(x[u][1][0], x[u][1][1]), hept_pack_bf16(x[u][1][2], x[u][1][3])};
            }
        }
    }
}

constexpr int BKT_BINS_PER_THREAD = LOBINS / BKT_THREADS;
static_assert(LOBINS % BKT_THREADS == 0, "every thread owns the same number of bins");
// MAXR = chunks whose run offsets fit the kernel's LDS table (64: every wave builds the table by itself with one
// shuffle scan, no barrier; the large-tile kernel takes up to 1024 chunks = 4 M keys per segment through a serial
// scan); longer segments walk the global table for every lookup -- slow, never wrong
// The common bucket of the REGION layout (nb <= LEAN_SLOTS * threads pairs, all of them in registers, low id bits embedded):
// the same three steps as the general code below -- histogram of the low id bits, exclusive prefix, scatter into the
// tile -- but every thread ranks ITS OWN pairs straight from its registers: the scatter's atomic told it the group, the
// bins array (now one past the last slot of every group) the group's bounds; a group of one or two pairs (92 % of the
// pairs at tracking-60k) is ranked by two tile reads and two compares, without a loop.  No validity masks, no
// run lookups, no second pass: ~45 % of the general path's instructions (the kernel is bound by instruction issue).
#ifndef HEPT_LEAN_SLOTS
#define HEPT_LEAN_SLOTS 4
#endif
constexpr int LEAN_SLOTS = HEPT_LEAN_SLOTS;
// pr[u] = pair u * threads + tid of the bucket for every u with u * threads < nb (anything where that position is >= nb)
__device__ __forceinline__ void bucket_lean(const unsigned long long (&pr)[LEAN_SLOTS], int nb, int start,
                                            unsigned long long* __restrict__ tile_s, unsigned int* __restrict__ cur_s,
                                            unsigned int* __restrict__ wsum_s, int* __restrict__ out) {
    static_assert(BKT_BINS_PER_THREAD == 4, "bins per thread: one 16-B LDS access");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned int* bin_s = cur_s + 1;
    unsigned int d[LEAN_SLOTS];
#pragma unroll
    for (int u = 0; u < LEAN_SLOTS; ++u) {
        if (u * BKT_THREADS >= nb) break;   // uniform
        d[u] = ((unsigned int)pr[u] >> EMBED_SHIFT) & (unsigned int)(LOBINS - 1);
        if (u * BKT_THREADS + tid < nb) atomicAdd(&bin_s[d[u]], 1u);
    }
    __syncthreads();
    {   // exclusive prefix over the bins: thread owns bins 4 tid .. 4 tid + 3
        const u32x4 c = *reinterpret_cast<const u32x4*>(bin_s + 4 * tid);
        const unsigned int tot = c[0] + c[1] + c[2] + c[3];
        const unsigned int incl = hept_wave_scan_add(tot);
        if (lane == 63) wsum_s[w] = incl;
        __syncthreads();
        unsigned int run = incl - tot;
#pragma unroll
        for (int ww = 0; ww < BKT_WAVES; ++ww)
            if (ww < w) run += wsum_s[ww];
        u32x4 e;
        e[0] = run; e[1] = run + c[0]; e[2] = e[1] + c[1]; e[3] = e[2] + c[2];
        *reinterpret_cast<u32x4*>(bin_s + 4 * tid) = e;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LEAN_SLOTS; ++u) {
        if (u * BKT_THREADS >= nb) break;   // uniform
        if (u * BKT_THREADS + tid < nb) tile_s[atomicAdd(&bin_s[d[u]], 1u)] = pr[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LEAN_SLOTS; ++u) {
        if (u * BKT_THREADS >= nb) break;   // uniform
        if (u * BKT_THREADS + tid < nb) {
            const unsigned long long p = pr[u];
            const int g0 = (int)cur_s[d[u]], g1 = (int)cur_s[d[u] + 1];   // the group: [end of the group before, own end)
            const unsigned long long a = tile_s[g0], b = tile_s[g1 - 1];   // first and last member (the same pair: alone)
            int smaller = (int)(a < p) + (int)(b < p);
            for (int j = g0 + 1; j < g1 - 1; ++j) smaller += (int)(tile_s[j] < p);   // groups of three and more
            out[start + g0 + smaller] = (int)((unsigned int)p & EMBED_INDEX_MASK);
        }
    }
}

// REGION (see RegionArgs): the bucket's pairs are ONE contiguous array, region[seg][bucket][0 .. nb), nb = cnt[seg][bucket],
// and its first final position is the sum of the counts in front of it -- no run table, no lookup, and nothing the pair
// loads depend on: the first two register slots are requested before the counts have arrived (a region is at least
// 2 * threads slots long; what lies behind nb is ignored).  A bucket of more than capb pairs first collects its pairs
// (the region + its share of the segment's overflow pool) into `gathered` and carries on from there.
// the LDS of one bucket (owned by the kernel that calls bucket_body)
struct BucketLds {
    unsigned long long* tile;   // [CAP]
    unsigned int* binarr;       // [LOBINS + 4] at a 16-B boundary: cur = binarr + 3 (cur[0] stays 0), bin d lives at cur[d + 1]
    unsigned int* wsum;         // [BKT_WAVES]
    unsigned int* roff;         // [MAXR + 1] pairs of the bucket in runs 0 .. c-1
    unsigned int* rbase;        // [MAXR + 1] first pair of run c, as an index into the segment's pairs; [MAXR]: start
    unsigned int* front;        // [BKT_WAVES + 1] REGION: per-wave sums of the counts in front; [BKT_WAVES]: gather cursor
};
template <int CAP, bool EMBED, int MAXR, bool REGION, bool LEAN>
__device__ __forceinline__ void bucket_body(const unsigned long long* __restrict__ pairs,
                                            unsigned long long* __restrict__ scratch,
                                            const SegParams* __restrict__ seg_params,
                                            const unsigned int* __restrict__ tab, int N, int n_chunks,
                                            int* __restrict__ pos_out, const RegionArgs& rga, const int seg, const int bucket,
                                            const BucketLds& lds) {
    static_assert(CAP >= LOBINS, "the tile also holds the splitters of the streaming path");
    unsigned long long* tile_s = lds.tile;
    unsigned int* cur_s = lds.binarr + 3;
    unsigned int* wsum_s = lds.wsum;
    unsigned int* roff_s = lds.roff;
    unsigned int* rbase_s = lds.rbase;
    unsigned int* front_s = lds.front;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long* seg_pairs = pairs + (size_t)seg * N;
    const unsigned int* btab = tab + (size_t)seg * n_chunks * TAB + bucket;   // + c * TAB: [first, end) of run c
    unsigned int* bin_s = cur_s + 1;
    auto zero_bins = [&]() {
#pragma unroll
        for (int u = 0; u < BKT_BINS_PER_THREAD; ++u) bin_s[u * BKT_THREADS + tid] = 0;
        if (tid == 0) cur_s[0] = 0;
    };
    zero_bins();
    // ---- the bucket's runs: offsets, first final position, size
    const bool table_in_lds = REGION || n_chunks <= MAXR;
    int start = 0, nb = 0, longest = 1 << 30;
    constexpr int SPEC = 2;                       // REGION: register slots requested before nb is known
    unsigned long long spec[SPEC] = {};
    const unsigned long long* src = nullptr;      // REGION: the bucket's pairs, contiguous
    if constexpr (REGION) {
        static_assert(NTOP % BKT_THREADS == 0, "every thread sums the same number of bucket counts");
        constexpr int CPT = NTOP / BKT_THREADS;
        const unsigned int* cseg = rga.cnt + (size_t)seg * NTOP;
        const unsigned long long* reg = rga.region + ((size_t)seg * NTOP + bucket) * rga.pitch;
#pragma unroll
        for (int u = 0; u < SPEC; ++u) spec[u] = hept_ld<HEPT_NT_KB_IN>(reg + u * BKT_THREADS + tid);
        unsigned int in_front = 0;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int b = tid * CPT + u;
            const unsigned int c = cseg[b];
            in_front += b < bucket ? c : 0u;
        }
        nb = (int)cseg[bucket];
        in_front = hept_wave_sum(in_front);
        if (lane == 0) front_s[w] = in_front;
        if (tid == 0) front_s[BKT_WAVES] = 0;
        src = reg;
    } else
    if (n_chunks <= HEPT_WAVE) {
        // every wave builds the table by itself (same values, written twice): no barrier before the lookups
        unsigned int a0 = 0, a1 = 0;
        if (lane < n_chunks) { a0 = btab[(size_t)lane * TAB]; a1 = btab[(size_t)lane * TAB + 1]; }
        const unsigned int incl = hept_wave_scan_add(a1 - a0);
        const unsigned int ssum = hept_wave_sum(a0);
        longest = (int)hept_wave_max(a1 - a0);
        if (lane < n_chunks) {
            roff_s[lane] = incl - (a1 - a0);
            rbase_s[lane] = (unsigned int)lane * SORT_CHUNK + a0;
        }
        nb = __builtin_amdgcn_readlane((int)incl, 63);
        start = (int)ssum;
        if (lane == 0) roff_s[n_chunks] = (unsigned int)nb;
    } else {
        if (table_in_lds) {
            for (int c = tid; c < n_chunks; c += BKT_THREADS) {
                const unsigned int a0 = btab[(size_t)c * TAB], a1 = btab[(size_t)c * TAB + 1];
                roff_s[c + 1] = a1 - a0;
                rbase_s[c] = (unsigned int)c * SORT_CHUNK + a0;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned int run = 0, ssum = 0;
                for (int c = 0; c < n_chunks; ++c) {
                    const unsigned int l = roff_s[c + 1];
                    ssum += rbase_s[c] - (unsigned int)c * SORT_CHUNK;
                    roff_s[c] = run;
                    run += l;
                }
                roff_s[n_chunks] = run;
                rbase_s[MAXR] = ssum;
            }
        } else if (tid == 0) {
            unsigned int run = 0, ssum = 0;
            for (int c = 0; c < n_chunks; ++c) {
                const unsigned int a0 = btab[(size_t)c * TAB], a1 = btab[(size_t)c * TAB + 1];
                ssum += a0;
                run += a1 - a0;
            }
            roff_s[0] = run;
            rbase_s[MAXR] = ssum;
        }
        __syncthreads();
        nb = (int)roff_s[table_in_lds ? n_chunks : 0];
        start = (int)rbase_s[MAXR];
    }
    // (the two sums come out of shuffles / LDS: tell the compiler that they are wave-uniform, or every quantity derived
    //  from them -- loop bounds, the path taken, the output base -- lives in vector registers and every branch on them
    //  becomes predicated code: 122 VGPRs instead of ~60)
    nb = __builtin_amdgcn_readfirstlane(nb);
    if (nb <= 0) return;   // uniform
    bool collected = false;   // REGION: the bucket overflowed its region and was collected into `gathered`
    if constexpr (REGION) {
        __syncthreads();   // the waves' sums of the counts in front (and the bins are zero in every wave's view)
#pragma unroll
        for (int ww = 0; ww < BKT_WAVES; ++ww) start += (int)front_s[ww];
#ifndef HEPT_BKT_NO_LEAN
        if constexpr (EMBED && LEAN) {
            if (nb <= LEAN_SLOTS * BKT_THREADS) {   // uniform: the common bucket
                unsigned long long pr[LEAN_SLOTS];
#pragma unroll
                for (int u = 0; u < LEAN_SLOTS; ++u) {
                    if (u >= SPEC && u * BKT_THREADS >= nb) break;   // uniform
                    pr[u] = u < SPEC ? spec[u < SPEC ? u : 0] : (u * BKT_THREADS + tid < nb ? src[u * BKT_THREADS + tid] : 0ull);
                }
                bucket_lean(pr, nb, __builtin_amdgcn_readfirstlane(start), tile_s, cur_s, wsum_s, pos_out + (size_t)seg * N);
                return;
            }
        }
#endif
        if (nb > (int)rga.capb) {   // uniform, uncommon: region + this bucket's share of the overflow pool -> gathered
            start = __builtin_amdgcn_readfirstlane(start);
            unsigned long long* gdst = rga.gathered + (size_t)seg * N + start;
            for (int i = tid; i < (int)rga.capb; i += BKT_THREADS) gdst[i] = src[i];
            const SegParams sp = seg_params[seg];
            const int n_pool = (int)rga.cnt[(size_t)rga.segs * NTOP + seg];
            const unsigned long long* pool = rga.pool + (size_t)seg * N;
            for (int j = tid; j < n_pool; j += BKT_THREADS) {
                const unsigned long long p = pool[j];
                if ((int)(bucket_id((unsigned int)(p >> 32), sp.kmin, sp.scale) >> TOP_SHIFT) == bucket)
                    gdst[rga.capb + atomicAdd(&front_s[BKT_WAVES], 1u)] = p;
            }
            __threadfence_block();
            __syncthreads();
            src = gdst;
            collected = true;
        }
    }
    start = __builtin_amdgcn_readfirstlane(start);
    longest = __builtin_amdgcn_readfirstlane(longest);
    // position in run order -> (run, first pair of the run, pairs in front of the run)
    struct Where { int c; unsigned int before, base, len; };
    auto locate = [&](unsigned int idx) -> Where {
        Where wv;
        if (table_in_lds) {
            int lo = 0, hi = n_chunks;   // the last c with roff[c] <= idx (runs of length 0 are passed over)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (roff_s[mid] <= idx) lo = mid; else hi = mid;
            }
            wv.c = lo;
            wv.before = roff_s[lo];
            wv.base = rbase_s[lo];
            wv.len = roff_s[lo + 1] - wv.before;
        } else {
            unsigned int before = 0;
            int c = 0;
            unsigned int a0 = 0, l = 0;
            for (; c < n_chunks; ++c) {
                a0 = btab[(size_t)c * TAB];
                l = btab[(size_t)c * TAB + 1] - a0;
                if (idx < before + l) break;
                before += l;
            }
            wv.c = c; wv.before = before; wv.base = (unsigned int)c * SORT_CHUNK + a0; wv.len = l;
        }
        return wv;
    };
    auto pair_at = [&](unsigned int idx) -> unsigned long long {
        if constexpr (REGION) return src[idx];
        const Where wv = locate(idx);
        return seg_pairs[(size_t)wv.base + (idx - wv.before)];
    };
    // ---- the first CAP pairs of the bucket: registers, every load in flight at once.  Two ways of dealing them out:
    //  * by run (one cloud: every chunk contributes a short run): LPR = threads / n_chunks lanes (a power of two) share
    //    run c = tid / LPR and take its pairs rl, rl + LPR, ...; no lookup at all.  Taken when every run fits its lanes'
    //    ITEMS register slots.
    //  * by position (a batch of clouds: a bucket draws hundreds of pairs from two or three chunks): position
    //    i = u * threads + tid in run order, one binary search over the run offsets per position; only
    //    ceil(nb / threads) of the ITEMS slots have an active lane, the others are skipped as a whole.
    constexpr int ITEMS = CAP / BKT_THREADS;
    const int n_reg = nb < CAP ? nb : CAP;
    int lpr_log = 0;
    while ((2 << lpr_log) * n_chunks <= BKT_THREADS) ++lpr_log;
#ifdef HEPT_BKT_NO_BY_RUN
    const bool by_run = false;
#else
    const bool by_run = !REGION && n_chunks <= HEPT_WAVE && n_chunks <= BKT_THREADS && longest <= (ITEMS << lpr_log);   // uniform
#endif
    unsigned long long mine[ITEMS];
    unsigned long long vmask = 0;   // bit u: slot u holds a pair (ITEMS <= 64)
    // slots in use anywhere in the workgroup (uniform): the passes below stop there instead of stepping through the
    // predicated-off code of the empty slots (the kernel is bound by VALU issue)
    const int n_slots = by_run ? min(ITEMS, (longest + (1 << lpr_log) - 1) >> lpr_log)
                               : (n_reg + BKT_THREADS - 1) / BKT_THREADS;
    if (by_run) {
        const int rc = tid >> lpr_log, rl = tid & ((1 << lpr_log) - 1);
        int rlen = 0;
        const unsigned long long* rsrc = seg_pairs;
        if (rc < n_chunks) {
            rlen = (int)(roff_s[rc + 1] - roff_s[rc]);
            rsrc = seg_pairs + rbase_s[rc];
        }
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
            const int j = rl + (u << lpr_log);
            mine[u] = j < rlen ? rsrc[j] : 0ull;
            vmask |= j < rlen ? 1ull << u : 0ull;
        }
    } else {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
            const int i = u * BKT_THREADS + tid;
            if (REGION && u < SPEC) mine[u] = i < n_reg ? (collected ? src[i] : spec[u]) : 0ull;
            else mine[u] = i < n_reg ? pair_at((unsigned int)i) : 0ull;
            vmask |= i < n_reg ? 1ull << u : 0ull;
        }
    }
    static_assert(ITEMS <= 64, "one validity bit per register slot");
    auto valid = [&](int u) { return (vmask >> u) & 1ull; };
    // every pair of the bucket that is NOT in a register (positions >= CAP: oversize buckets only)
    auto for_rest = [&](auto&& f) {
        for (int i = CAP + tid; i < nb; i += BKT_THREADS) f(pair_at((unsigned int)i));
    };
    int* out = pos_out + (size_t)seg * N + start;
    const SegParams rg = seg_params[seg];
    bool in_lds = nb <= 2 * CAP;                 // workgroup-uniform
    unsigned long long* spl_s = tile_s;          // the streaming path groups in global scratch, its tile is free
    auto lo_of = [&](unsigned long long p) -> unsigned int {
        if (in_lds)
            return EMBED ? ((unsigned int)p >> EMBED_SHIFT) & (unsigned int)(LOBINS - 1)
                         : bucket_id((unsigned int)(p >> 32), rg.kmin, rg.scale) & (unsigned int)(LOBINS - 1);
        int lo = 0, hi = LOBINS;  // number of splitters <= p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (spl_s[mid] <= p) lo = mid + 1; else hi = mid;
        }
        return (unsigned int)(lo > 0 ? lo - 1 : 0);
    };
    auto prefix_bins = [&]() {  // exclusive prefix over the bins: thread owns bins BPT*tid .. BPT*tid + BPT - 1
        __syncthreads();   // the histogram is complete
        unsigned int c[BKT_BINS_PER_THREAD], tot = 0;
#pragma unroll
        for (int u = 0; u < BKT_BINS_PER_THREAD; ++u) { c[u] = bin_s[BKT_BINS_PER_THREAD * tid + u]; tot += c[u]; }
        const unsigned int incl = hept_wave_scan_add(tot);
        if (lane == 63) wsum_s[w] = incl;
        __syncthreads();
        unsigned int run = incl - tot;
#pragma unroll
        for (int ww = 0; ww < BKT_WAVES; ++ww)
            if (ww < w) run += wsum_s[ww];
#pragma unroll
        for (int u = 0; u < BKT_BINS_PER_THREAD; ++u) {
            bin_s[BKT_BINS_PER_THREAD * tid + u] = run;
            run += c[u];
        }
        __syncthreads();
    };
    if constexpr (!REGION) __syncthreads();   // the bins are zero in every wave's view (REGION: the barrier of the front sums)
    int n_low = nb;  // pairs in the lower half of the id bins (two-pass buckets)
    if (in_lds) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
            if (valid(u)) atomicAdd(&bin_s[lo_of(mine[u])], 1u);
        }
        for_rest([&](unsigned long long p) { atomicAdd(&bin_s[lo_of(p)], 1u); });
        prefix_bins();
        if (nb > CAP) {
            n_low = (int)bin_s[LOBINS / 2];
            if (n_low > CAP || nb - n_low > CAP) in_lds = false;  // uniform: every thread reads the same word
        }
    }
    if (!in_lds) {
        // splitters: pairs at regular positions of the bucket in run order
        unsigned long long sample[LOBINS / BKT_THREADS];
#pragma unroll
        for (int u = 0; u < LOBINS / BKT_THREADS; ++u)
            sample[u] = pair_at((unsigned int)((size_t)(u * BKT_THREADS + tid) * nb / LOBINS));
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LOBINS / BKT_THREADS; ++u) spl_s[u * BKT_THREADS + tid] = sample[u];
        zero_bins();
        __syncthreads();
        for (int k = 2; k <= LOBINS; k <<= 1)          // bitonic sort, ascending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < LOBINS; i += BKT_THREADS) {
                    const int partner = i ^ j;
                    if (partner > i) {
                        const unsigned long long a = spl_s[i], b = spl_s[partner];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { spl_s[i] = b; spl_s[partner] = a; }
                    }
                }
                __syncthreads();
            }
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
            if (valid(u)) atomicAdd(&bin_s[lo_of(mine[u])], 1u);
        }
        for_rest([&](unsigned long long p) { atomicAdd(&bin_s[lo_of(p)], 1u); });
        prefix_bins();
    }
    // rank the pairs grouped[0 .. cnt) = bucket positions [off, off + cnt) inside their groups
    // (separate LDS / global code paths: one pointer for both would compile to slow FLAT accesses)
    auto rank_all = [&](const unsigned long long* grouped, int off, int cnt) {
        for (int i = tid; i < cnt; i += BKT_THREADS) {
            const unsigned long long p = grouped[i];
            const unsigned int d = lo_of(p);
            const int g0 = (int)cur_s[d] - off, g1 = (int)cur_s[d + 1] - off;  // = bin_s[d - 1], bin_s[d]
            int smaller = 0;
            // (a pair alone in its id group -- most of them -- needs no look at the tile: the kernel is bound by LDS
            //  instruction throughput, every skipped read counts)
            for (int j = g0; g1 - g0 > 1 && j < g1; j += 4) {  // 4 independent reads per round trip
                unsigned long long q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = grouped[min(j + u, g1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) smaller += (j + u < g1) && (q[u] < p);
            }
            out[off + g0 + smaller] = (int)(EMBED ? (unsigned int)p & EMBED_INDEX_MASK : (unsigned int)p);
        }
    };
    if (in_lds) {
        const int n_pass = nb > CAP ? 2 : 1;
        for (int ps = 0; ps < n_pass; ++ps) {
            const unsigned int half = (unsigned int)ps;                  // bins [0, LOBINS/2) then [LOBINS/2, LOBINS)
            const int off = ps ? n_low : 0, cnt = n_pass == 2 ? (ps ? nb - n_low : n_low) : nb;
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {
                if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
                if (valid(u)) {
                    const unsigned int d = lo_of(mine[u]);
                    if (n_pass == 1 || (d >> (TOP_SHIFT - 1)) == half) tile_s[atomicAdd(&bin_s[d], 1u) - off] = mine[u];
                }
            }
            for_rest([&](unsigned long long p) {
                const unsigned int d = lo_of(p);
                if (n_pass == 1 || (d >> (TOP_SHIFT - 1)) == half) tile_s[atomicAdd(&bin_s[d], 1u) - off] = p;
            });
            __syncthreads();
            rank_all(tile_s, off, cnt);
            __syncthreads();
        }
    } else {
        unsigned long long* g = scratch + (size_t)seg * N + start;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            if (ITEMS <= 16 && u >= n_slots) break;   // (longer register tiles: the early exit would keep the loop from unrolling)
            if (valid(u)) g[atomicAdd(&bin_s[lo_of(mine[u])], 1u)] = mine[u];
        }
        for_rest([&](unsigned long long p) { g[atomicAdd(&bin_s[lo_of(p)], 1u)] = p; });
        __threadfence_block();
        __syncthreads();
        rank_all(g, 0, nb);
    }
}

template <int CAP, bool EMBED, int MAXR, bool REGION>
__global__ __launch_bounds__(BKT_THREADS) void bucket_sort_kernel(const unsigned long long* __restrict__ pairs,
                                                                  unsigned long long* __restrict__ scratch,
                                                                  const SegParams* __restrict__ seg_params,
                                                                  const unsigned int* __restrict__ tab, int N,
                                                                  int n_chunks, int* __restrict__ pos_out,
                                                                  RowsJob rows, RegionArgs rga) {
    if ((int)blockIdx.y < rows.vy) {   // uniform: a rider workgroup (the first rows of the grid: dispatched first)
        const unsigned int wg = blockIdx.y * gridDim.x + blockIdx.x, n_wgs = (unsigned int)rows.vy * gridDim.x;
        if (rows.H == 8) rows_rider<8>(rows, wg, n_wgs);
        else rows_rider<0>(rows, wg, n_wgs);
        return;
    }
#ifdef HEPT_RIDERS_ONLY   // measurement builds: what the riders cost by themselves
    if (rows.vy > 0) return;
#endif
    __shared__ unsigned long long tile_s[CAP];
    __shared__ __attribute__((aligned(16))) unsigned int binarr_s[LOBINS + 4];
    __shared__ unsigned int wsum_s[BKT_WAVES];
    __shared__ unsigned int roff_s[REGION ? 1 : MAXR + 1];
    __shared__ unsigned int rbase_s[REGION ? 1 : MAXR + 1];
    __shared__ unsigned int front_s[REGION ? BKT_WAVES + 1 : 1];
    const BucketLds lds{tile_s, binarr_s, wsum_s, roff_s, rbase_s, front_s};
    bucket_body<CAP, EMBED, MAXR, REGION, true>(pairs, scratch, seg_params, tab, N, n_chunks, pos_out, rga,
                                                (int)blockIdx.y - rows.vy, (int)blockIdx.x, lds);
}

// src variant: per (table, head) upper bound of the shift in units of span: max_n (phi * cfac + eta), written into
// the third column of partial slot 0 (the prep kernel left 0 there), so that K1 bounds the key range with it
__global__ __launch_bounds__(SORT_THREADS) void src_bound_kernel(const float* __restrict__ eta_idx,
                                                                 const float* __restrict__ phi_idx,
                                                                 const float* __restrict__ cfac, int N, int H, int t0,
                                                                 float* __restrict__ minmax) {
    __shared__ float red_s[SORT_WAVES];
    const int th = blockIdx.x, t = th / H, h = th % H, tid = threadIdx.x;
    const size_t row_off = ((size_t)(t0 + t) * H + h) * N;
    const float cf = cfac[(size_t)(t0 + t) * H + h];
    float m = 0.f;
    for (int n = tid; n < N; n += SORT_THREADS) {
        const float e = eta_idx[row_off + n], p = phi_idx[row_off + n];
        if (e < INFINITY && p < INFINITY) m = fmaxf(m, fmaf(p, cf, e));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) red_s[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
        minmax[(((size_t)t * H + h) * HEPT_PREP_GRID + 0) * 4 + 2] = m * 1.0001f + 1.f;  // slack for the roundings
    }
}

// ---- generic front end (hept_segmented_argsort): S segments of L raw fp32 keys, +inf allowed as padding ----
// finite min / max of every segment: one workgroup per 4096-key chunk folds into two ordered-uint words per segment
// with one atomicMin each (word 0: min, word 1: complement of the max; both start at 0xFFFFFFFF)
__global__ __launch_bounds__(SORT_THREADS) void raw_range_kernel(const float* __restrict__ keys, int L,
                                                                 const int* __restrict__ seg_len,
                                                                 unsigned int* __restrict__ range_bits) {
    __shared__ unsigned int red_s[2][SORT_WAVES];
    const int tid = threadIdx.x, seg = blockIdx.y;
    const float* k = keys + (size_t)seg * L;
    const int len = seg_len ? seg_len[seg] : L;
    unsigned int lo = 0xFFFFFFFFu, nhi = 0xFFFFFFFFu;
    const int base = blockIdx.x * SORT_CHUNK;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < len) {
            const float x = k[n];
            if (x < INFINITY && x > -INFINITY) {
                const unsigned int u = ordered_bits(x);
                lo = min(lo, u);
                nhi = min(nhi, ~u);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, (unsigned int)__shfl_xor((int)lo, off));
        nhi = min(nhi, (unsigned int)__shfl_xor((int)nhi, off));
    }
    if ((tid & 63) == 0) { red_s[0][tid >> 6] = lo; red_s[1][tid >> 6] = nhi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int ww = 1; ww < SORT_WAVES; ++ww) { lo = min(lo, red_s[0][ww]); nhi = min(nhi, red_s[1][ww]); }
        if (lo != 0xFFFFFFFFu) atomicMin(range_bits + 2 * seg, lo);
        if (nhi != 0xFFFFFFFFu) atomicMin(range_bits + 2 * seg + 1, nhi);
    }
}

// ---- short segments (N <= SMALL_CAP: the 4k / 6k clouds): the whole sort of a segment in ONE workgroup ----------------
// Three dependent kernels of ~5-10 us each are a fixed ~23 us at these sizes; here a 1024-thread workgroup keeps the
// segment's (key, index) pairs in registers, groups them by a 12-bit monotone id in LDS (histogram / prefix / scatter on
// one bins array, as in K4) and ranks every pair inside its id group.  Same exact result.
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_WAVES = SMALL_THREADS / HEPT_WAVE;
constexpr int SMALL_CAP = 6144;
constexpr int SMALL_ITEMS = SMALL_CAP / SMALL_THREADS;
constexpr int SMALL_BINS = 4096;
constexpr size_t SMALL_LDS = (size_t)SMALL_CAP * 8 + (SMALL_BINS + 1) * 4;

// Round 5 -- Q workgroups per segment, split by id range (SMALL_SPLIT = 4): one workgroup per segment kept 48 of the 256
// CUs busy for 13.7 us at tracking-6k, and most of that is the per-thread chain of LDS round trips of the scatter and
// rank phases over six keys per thread.  Workgroup q of a segment forms ALL keys (the loads are the cheap part) but keeps
// only those whose id falls into its quarter of the id range: it counts the keys below its range (their number is where
// its own positions start), histograms / prefixes / scatters / ranks its own ~N/Q keys -- a quarter of the bins per
// thread, a third of the rank slots.  Any split of a monotone id range gives the exact sort; a skewed one (all keys in one
// quarter) degrades to the one-workgroup time.
#ifndef HEPT_SMALL_SPLIT
#define HEPT_SMALL_SPLIT 4
#endif
constexpr int SMALL_SPLIT = HEPT_SMALL_SPLIT;
static_assert(SMALL_SPLIT >= 1 && SMALL_BINS % SMALL_SPLIT == 0 && (SMALL_BINS / SMALL_SPLIT) % SMALL_THREADS == 0,
              "every thread owns the same number of a workgroup's bins");
// MODE 0: key = proj + float(code) * span;  MODE 1: src variant (get_geo_shift);  MODE 2: raw keys (S segments of L)
template <int MODE, int Q>
__global__ __launch_bounds__(SMALL_THREADS) void small_sort_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ eta_idx, const float* __restrict__ phi_idx, const float* __restrict__ cfac,
    const float* __restrict__ minmax, int N_stride, int H, int t0, int Tl, int* __restrict__ pos_out,
    const int* __restrict__ seg_len) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long small_tile_s[];  // [SMALL_CAP] pairs, then bins
    __shared__ float red_s[3][SMALL_WAVES];
    __shared__ unsigned int wsum_s[SMALL_WAVES];
    __shared__ unsigned int below_s[SMALL_WAVES];
    unsigned int* cur_s = reinterpret_cast<unsigned int*>(small_tile_s + SMALL_CAP);  // [0] stays 0; bin d at [d + 1]
    unsigned int* bin_s = cur_s + 1;
    constexpr int BQ = SMALL_BINS / Q;   // bins of this workgroup: ids [part * BQ, (part + 1) * BQ)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, seg = blockIdx.x / Q, part = blockIdx.x % Q;
    const int N = seg_len ? seg_len[seg] : N_stride;  // keys that take part (ragged argsort); N_stride = segment pitch
    for (int i = tid; i < BQ + 1; i += SMALL_THREADS) cur_s[i] = 0;

    // MODE 0: the hashes and codes of this thread's keys are requested first -- they do not depend on the key range,
    // and the launch is one workgroup per segment: its length is this chain of dependent loads
    float pj0[SMALL_ITEMS];
    long long cd0[SMALL_ITEMS];
    if (MODE == 0) {
        const int th = seg % (Tl * H);
        const float* pr = (seg >= Tl * H ? kproj : qproj) + (size_t)th * N_stride;
        const int64_t* cr = codes + ((size_t)(t0 + th / H) * H + th % H) * N_stride;
#pragma unroll
        for (int u = 0; u < SMALL_ITEMS; ++u) {
            const int n = u * SMALL_THREADS + tid;
            pj0[u] = n < N ? pr[n] : 0.f;
            cd0[u] = n < N ? cr[n] : 0;
        }
    }

    // ---- key range of the segment
    float lo = INFINITY, hi = -INFINITY, cmax = 0.f;
    const float* proj;
    size_t row_off = 0;
    if (MODE == 2) {
        proj = qproj + (size_t)seg * N_stride;  // raw keys
        for (int i = tid; i < N; i += SMALL_THREADS) {
            const float x = proj[i];
            if (x < INFINITY && x > -INFINITY) { lo = fminf(lo, x); hi = fmaxf(hi, x); }
        }
    } else {
        const int th = seg % (Tl * H), t = th / H, h = th % H;
        proj = (seg >= Tl * H ? kproj : qproj) + (size_t)th * N_stride;
        row_off = ((size_t)(t0 + t) * H + h) * N_stride;
        for (int i = tid; i < HEPT_PREP_GRID; i += SMALL_THREADS) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(minmax + (((size_t)t * H + h) * HEPT_PREP_GRID + i) * 4);
            lo = fminf(lo, m[0]); hi = fmaxf(hi, m[1]); cmax = fmaxf(cmax, m[2]);
        }
        if (MODE == 1) {  // src variant: bound of the shift in units of span (what src_bound_kernel computes)
            const float cf = cfac[(size_t)(t0 + t) * H + h];
            float m = 0.f;
            for (int i = tid; i < N; i += SMALL_THREADS) {
                const float e = eta_idx[row_off + i], p = phi_idx[row_off + i];
                if (e < INFINITY && p < INFINITY) m = fmaxf(m, fmaf(p, cf, e));
            }
            cmax = m * 1.0001f + 1.f;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
        cmax = fmaxf(cmax, __shfl_xor(cmax, off));
    }
    if (lane == 0) { red_s[0][w] = lo; red_s[1][w] = hi; red_s[2][w] = cmax; }
    __syncthreads();
    lo = red_s[0][0]; hi = red_s[1][0]; cmax = red_s[2][0];
#pragma unroll
    for (int ww = 1; ww < SMALL_WAVES; ++ww) {
        lo = fminf(lo, red_s[0][ww]); hi = fmaxf(hi, red_s[1][ww]); cmax = fmaxf(cmax, red_s[2][ww]);
    }
    if (lo > hi) { lo = 0.f; hi = 0.f; }  // no finite key at all
    const float span = hi - lo;
    const float width = MODE == 2 ? span : (hi + cmax * span) - lo;
    float scale = width > 0.f ? (float)SMALL_BINS / width : 0.f;
    if (!(scale < 3.0e38f)) scale = 0.f;
    auto bin_of = [&](unsigned int u) -> unsigned int {
        const float x = fminf((from_ordered(u) - lo) * scale, (float)(SMALL_BINS - 1));
        const int b = (int)x;
        return (unsigned int)(b < 0 ? 0 : b);
    };

    // ---- keys -> pairs in registers, histogram of the id
    unsigned long long mine[SMALL_ITEMS];
    unsigned int below = 0;   // keys of this thread whose id lies below the workgroup's range
    float cf = 0.f;
    if (MODE == 1) { const int th = seg % (Tl * H); cf = cfac[(size_t)(t0 + th / H) * H + th % H]; }
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {
        const int n = u * SMALL_THREADS + tid;
        mine[u] = ~0ull;   // (not a pair of this workgroup)
        if (n < N) {
            float key;
            if (MODE == 0) {
                float off = (float)cd0[u] * span;   // two separately rounded ops, as in K1
                asm volatile("" : "+v"(off));
                key = pj0[u] + off;
            } else if (MODE == 1) {
                float t1 = eta_idx[row_off + n] * span;
                asm volatile("" : "+v"(t1));
                float t2 = phi_idx[row_off + n] * span;
                asm volatile("" : "+v"(t2));
                t2 = t2 * cf;
                asm volatile("" : "+v"(t2));
                float t3 = t2 + t1;
                asm volatile("" : "+v"(t3));
                key = proj[n] + t3;
            } else {
                key = proj[n];
            }
            const unsigned int ub = ordered_bits(key);
            const unsigned int b = bin_of(ub);
            if (Q == 1 || b / BQ == (unsigned int)part) {
                mine[u] = ((unsigned long long)ub << 32) | (unsigned int)n;
                atomicAdd(&bin_s[b - part * BQ], 1u);
            } else {
                below += b < (unsigned int)(part * BQ) ? 1u : 0u;
            }
        }
    }
    if (Q > 1) {
        below = hept_wave_sum(below);
        if (lane == 0) below_s[w] = below;
    }
    __syncthreads();
    unsigned int n_mine;   // pairs of this workgroup
    {   // exclusive prefix over the bins: thread owns BPT consecutive bins
        constexpr int BPT = BQ / SMALL_THREADS;
        unsigned int c[BPT], tot = 0;
#pragma unroll
        for (int u = 0; u < BPT; ++u) { c[u] = bin_s[BPT * tid + u]; tot += c[u]; }
        const unsigned int incl = hept_wave_scan_add(tot);
        if (lane == 63) wsum_s[w] = incl;
        __syncthreads();
        unsigned int run = incl - tot;
        n_mine = 0;
        below = 0;
#pragma unroll
        for (int ww = 0; ww < SMALL_WAVES; ++ww) {
            const unsigned int ws = wsum_s[ww];
            run += ww < w ? ws : 0u;
            n_mine += ws;
            if (Q > 1) below += below_s[ww];
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            bin_s[BPT * tid + u] = run;
            run += c[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u)
        if (mine[u] != ~0ull)
            small_tile_s[atomicAdd(&bin_s[bin_of((unsigned int)(mine[u] >> 32)) - part * BQ], 1u)] = mine[u];
    __syncthreads();
    n_mine = (unsigned int)__builtin_amdgcn_readfirstlane((int)n_mine);
    const int n_slots = ((int)n_mine + SMALL_THREADS - 1) / SMALL_THREADS;   // rank slots in use (uniform): ~N / (Q * threads)
    int* out = pos_out + (size_t)seg * N_stride + below;
    // rank inside the id group.  The launch is one workgroup per segment, so its length is this thread's chain: the
    // SMALL_ITEMS positions of a thread are independent -- their tile reads, group bounds and first group rounds are
    // issued side by side (a loop over them was SMALL_ITEMS dependent chains of four LDS round trips each)
    unsigned long long pv[SMALL_ITEMS];
    int g0v[SMALL_ITEMS], g1v[SMALL_ITEMS], smv[SMALL_ITEMS];
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {
        if (u >= n_slots) break;
        const int i = u * SMALL_THREADS + tid;
        pv[u] = small_tile_s[i < (int)n_mine ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {
        if (u >= n_slots) break;
        const unsigned int d = bin_of((unsigned int)(pv[u] >> 32)) - part * BQ;
        g0v[u] = (int)cur_s[d];
        g1v[u] = (int)cur_s[d + 1];
    }
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {   // first round of every group: up to 4 members
        if (u >= n_slots) break;
        unsigned long long q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = small_tile_s[min(g0v[u] + e, g1v[u] - 1)];
        int sm = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) sm += (g0v[u] + e < g1v[u]) && (q[e] < pv[u]);
        smv[u] = sm;
    }
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {
        if (u >= n_slots) break;
        const int i = u * SMALL_THREADS + tid;
        int smaller = smv[u];
        for (int j = g0v[u] + 4; j < g1v[u]; j += 4) {   // larger groups: further rounds
            unsigned long long q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = small_tile_s[min(j + e, g1v[u] - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) smaller += (j + e < g1v[u]) && (q[e] < pv[u]);
        }
        if (i < (int)n_mine) out[g0v[u] + smaller] = (int)(unsigned int)pv[u];
    }
}

template <int MODE>
int launch_small_sort(int segs, hipStream_t st, const float* qproj, const float* kproj, const int64_t* codes,
                      const float* eta, const float* phi, const float* cfac, const float* minmax, int N, int H, int t0,
                      int Tl, int* pos, const int* seg_len = nullptr) {
    // one workgroup per segment when the launch already fills the chip, or when the segments are too short to be worth
    // splitting (HEPT_SMALL_NO_SPLIT=1: always; A/B runs)
    static const bool no_split = [] { const char* e = getenv("HEPT_SMALL_NO_SPLIT"); return e && *e && *e != '0'; }();
    if (SMALL_SPLIT > 1 && !no_split && segs <= 128 && N > SMALL_THREADS) {
        static LdsRaised raised;
        if (hept_raise_lds(raised, reinterpret_cast<const void*>(small_sort_kernel<MODE, SMALL_SPLIT>), SMALL_LDS)) return HEPT_ERR_LAUNCH;
        hipLaunchKernelGGL((small_sort_kernel<MODE, SMALL_SPLIT>), dim3(segs * SMALL_SPLIT), dim3(SMALL_THREADS), SMALL_LDS, st,
                           qproj, kproj, codes, eta, phi, cfac, minmax, N, H, t0, Tl, pos, seg_len);
        return hept_launch_status();
    }
    static LdsRaised raised1;
    if (hept_raise_lds(raised1, reinterpret_cast<const void*>(small_sort_kernel<MODE, 1>), SMALL_LDS)) return HEPT_ERR_LAUNCH;
    hipLaunchKernelGGL((small_sort_kernel<MODE, 1>), dim3(segs), dim3(SMALL_THREADS), SMALL_LDS, st, qproj, kproj, codes, eta,
                       phi, cfac, minmax, N, H, t0, Tl, pos, seg_len);
    return hept_launch_status();
}

// the two passes shared by hept_sort_tables and hept_segmented_argsort
struct SortBuffers {
    unsigned long long *pa, *pb;   // pairs (chunk runs; REGION: the overflow pools), scratch of the streaming path
    unsigned int* tab;             // [segs][n_chunks][257] digit offsets of every chunk's run
    unsigned int* range;           // [segs][2] ordered-uint finite min / ~max of raw keys (hept_segmented_argsort)
    SegParams* params;
    RegionArgs rg;                 // REGION layout (rg.region == nullptr: segments outside its range)
    size_t zero_bytes;             // bytes at rg.cnt that must be zero when KA starts
    size_t bytes;
};
inline bool region_sort_off() {   // HEPT_SORT_LINEAR=1: the round-3 layout (chunk runs + run tables) for A/B runs; read once
    static const bool off = [] { const char* e = getenv("HEPT_SORT_LINEAR"); return e && *e && *e != '0'; }();
    return off;
}
inline bool region_range(int N) { return N > SMALL_CAP && N <= REGION_MAX_N; }
SortBuffers carve_sort(void* sort_ws, int segs, int N) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    char* ws = reinterpret_cast<char*>(sort_ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* r = ws ? ws + off : nullptr;
        off += al(bytes);
        return r;
    };
    SortBuffers b{};
    b.pa = reinterpret_cast<unsigned long long*>(take((size_t)segs * N * 8));
    b.pb = reinterpret_cast<unsigned long long*>(take((size_t)segs * N * 8));
    b.tab = reinterpret_cast<unsigned int*>(take((size_t)segs * n_chunks * TAB * 4));
    b.range = reinterpret_cast<unsigned int*>(take((size_t)segs * 8));
    b.params = reinterpret_cast<SegParams*>(take((size_t)segs * sizeof(SegParams)));
    if (region_range(N)) {
        b.rg.capb = region_cap(N);
        b.rg.pitch = region_pitch(b.rg.capb);
        b.rg.segs = segs;
        b.zero_bytes = ((size_t)segs * NTOP + segs) * 4;
        b.rg.cnt = reinterpret_cast<unsigned int*>(take(b.zero_bytes));
        b.rg.gathered = reinterpret_cast<unsigned long long*>(take((size_t)segs * N * 8));
        b.rg.region = reinterpret_cast<unsigned long long*>(take((size_t)segs * NTOP * b.rg.pitch * 8));
        b.rg.pool = b.pa;
    }
    b.bytes = off;
    return b;
}
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* __restrict__ p, unsigned int words) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i < words) p[i] = 0u;
}
constexpr int BKT_CAP_SMALL = HEPT_BKT_CAP;  // LDS tile: the average bucket is N/NTOP
constexpr int BKT_CAP_LARGE = 6 * HEPT_BKT_CAP;  // 48 KiB tile for longer segments (average bucket up to ~3000 pairs)
static_assert(REGION_MAX_N == NTOP * (BKT_CAP_SMALL / 2), "the REGION layout serves exactly the small-tile bucket kernel");
// `zeroed`: the caller has already zeroed b.zero_bytes at b.rg.cnt on this stream (the row builder of the same forward)
template <int MODE, bool EMBED>
int run_passes_impl(const SortBuffers& b, int segs, int N, int* pos, hipStream_t st, const float* qproj,
                    const float* kproj, const int64_t* codes, const float* eta, const float* phi, const float* cfac,
                    const float* minmax, int H, int t0, int Tl, const int* seg_len, const float* bounds,
                    const RowsJob& rows, bool zeroed) {
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const dim3 grid4(NTOP, segs + rows.vy);   // rider rows first
    if (b.rg.region && !region_sort_off()) {
        // (a caller whose preceding kernel did not clear the counters: a 48-workgroup kernel of our own -- the runtime's
        //  fill kernel took 4.9 us for these 49 KB)
        if (!zeroed) {
            const unsigned int words = (unsigned int)(b.zero_bytes / 4);
            hipLaunchKernelGGL(zero_words_kernel, dim3((words + 255) / 256), dim3(256), 0, st, b.rg.cnt, words);
        }
        hipLaunchKernelGGL((chunk_sort_kernel<MODE, EMBED, true>), dim3(n_chunks, segs), dim3(SCT), 0, st, qproj, kproj, codes,
                           eta, phi, cfac, minmax, bounds ? nullptr : b.range, bounds ? bounds[0] : 0.f,
                           bounds ? bounds[1] : 0.f, N, H, t0, Tl, n_chunks, seg_len, b.params, b.pa, b.tab, b.rg);
        if (MODE != 2) hept_prof_mark_sort_mid(st);
        hipLaunchKernelGGL((bucket_sort_kernel<BKT_CAP_SMALL, EMBED, 64, true>), grid4, dim3(BKT_THREADS), 0, st, b.pa, b.pb,
                               b.params, b.tab, N, n_chunks, pos, rows, b.rg);
        return HEPT_OK;
    }
    hipLaunchKernelGGL((chunk_sort_kernel<MODE, EMBED, false>), dim3(n_chunks, segs), dim3(SCT), 0, st, qproj, kproj, codes, eta,
                       phi, cfac, minmax, bounds ? nullptr : b.range, bounds ? bounds[0] : 0.f, bounds ? bounds[1] : 0.f, N,
                       H, t0, Tl, n_chunks, seg_len, b.params, b.pa, b.tab, RegionArgs{});
    if (MODE != 2) hept_prof_mark_sort_mid(st);
    if ((size_t)N <= (size_t)NTOP * (BKT_CAP_SMALL / 2))
        hipLaunchKernelGGL((bucket_sort_kernel<BKT_CAP_SMALL, EMBED, 64, false>), grid4, dim3(BKT_THREADS), 0, st, b.pa, b.pb, b.params,
                           b.tab, N, n_chunks, pos, rows, RegionArgs{});
    else
        hipLaunchKernelGGL((bucket_sort_kernel<BKT_CAP_LARGE, EMBED, 1024, false>), grid4, dim3(BKT_THREADS), 0, st, b.pa, b.pb,
                           b.params, b.tab, N, n_chunks, pos, rows, RegionArgs{});
    return HEPT_OK;
}
template <int MODE>
int run_passes(const SortBuffers& b, int segs, int N, int* pos, hipStream_t st, const float* qproj, const float* kproj,
               const int64_t* codes, const float* eta, const float* phi, const float* cfac, const float* minmax, int H,
               int t0, int Tl, const int* seg_len = nullptr, const float* bounds = nullptr,
               const RowsJob& rows = RowsJob{}, bool zeroed = false) {
    if (N <= (1 << EMBED_SHIFT))
        return run_passes_impl<MODE, true>(b, segs, N, pos, st, qproj, kproj, codes, eta, phi, cfac, minmax, H, t0, Tl, seg_len,
                                           bounds, rows, zeroed);
    return run_passes_impl<MODE, false>(b, segs, N, pos, st, qproj, kproj, codes, eta, phi, cfac, minmax, H, t0, Tl, seg_len,
                                        bounds, rows, zeroed);
}

// the job of the rider workgroups, from the description a caller in another translation unit hands over
RowsJob rows_job(const HeptRowsJob* r) {
    RowsJob jb{};
    if (!r) return jb;
    jb.v = r->v;
    jb.kvhat = reinterpret_cast<char*>(r->kvhat);
    jb.N = r->N;
    jb.raw_size = r->raw_size;
    jb.H = r->H;
    jb.D4 = r->D / 4;
    jb.f32 = (r->precision == HEPT_PREC_F32 || r->precision == HEPT_PREC_F32_MFMA || r->precision == HEPT_PREC_F32_DIFF) ? 1 : 0;
    static const int vy = [] { const char* e = getenv("HEPT_ROW_RIDERS"); const int n = e ? atoi(e) : 0; return n > 0 && n <= 64 ? n : HEPT_ROWS_RIDERS; }();
    jb.vy = vy;   // (HEPT_ROW_RIDERS=<grid rows>: tuning; read once)
    return jb;
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_bytes(size_t segs, size_t N) { return carve_sort(nullptr, (int)segs, (int)N).bytes; }

extern "C" size_t hept_sort_workspace_bytes(int N, int H, int Tl) { return sort_bytes((size_t)2 * Tl * H, N); }

// internal (common.h): the block of `sort_ws` (for hept_sort_tables*_rows of 2*Tl*H segments of N keys) that has to be zero
// when the sort starts -- *bytes == 0: none.  A caller whose preceding kernel zeroes it passes zeroed = true.
void hept_sort_zero_block(void* sort_ws, int N, int H, int Tl, void** ptr, size_t* bytes) {
    *ptr = nullptr;
    *bytes = 0;
    if (!sort_ws || N <= SMALL_CAP || region_sort_off()) return;
    const SortBuffers b = carve_sort(sort_ws, 2 * Tl * H, N);
    if (!b.rg.region) return;
    *ptr = b.rg.cnt;
    *bytes = b.zero_bytes;
}

// internal (common.h): the sort of N-key segments is the KA / KB pair, whose KB launch can carry the v rows
// (segments of the small-tile bucket kernel only: the large-tile one runs two waves per SIMD at 218 VGPRs, and riders
//  in it cost 1.5 % at 480k and 1.9M points instead of saving anything)
bool hept_sort_carries_rows(int N, int H, int D) {
    return N > SMALL_CAP && (size_t)N <= (size_t)NTOP * (BKT_CAP_SMALL / 2) && D >= 4 && D % 4 == 0 && D <= 28 && H >= 1;
}

extern "C" int hept_sort_tables(const float* qproj, const float* kproj, const int64_t* codes, const float* minmax,
                                int N, int H, int T, int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos,
                                void* stream) {
    return hept_sort_tables_rows(qproj, kproj, codes, minmax, N, H, T, t0, Tl, sort_ws, qpos, kpos, nullptr, stream);
}

// internal (common.h): hept_sort_tables whose bucket-sort launch also writes the v rows described by `rows` (or null)
int hept_sort_tables_rows(const float* qproj, const float* kproj, const int64_t* codes, const float* minmax, int N, int H,
                          int T, int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos, const HeptRowsJob* rows,
                          void* stream, bool zeroed) {
    if (!qproj || !kproj || !codes || !minmax || !sort_ws || !qpos || !kpos) return HEPT_ERR_ARG;
    if (rows && (!rows->v || !rows->kvhat || rows->N != N || !hept_sort_carries_rows(N, rows->H, rows->D))) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (kpos != qpos + (size_t)Tl * H * N) return HEPT_ERR_ARG;  // one (2,Tl,H,N) array: q then k
    hipStream_t st = (hipStream_t)stream;
    const int segs = 2 * Tl * H;
    if (N <= SMALL_CAP)
        return launch_small_sort<0>(segs, st, qproj, kproj, codes, nullptr, nullptr, nullptr, minmax, N, H, t0, Tl, qpos);
    const SortBuffers b = carve_sort(sort_ws, segs, N);
    const int rc = run_passes<0>(b, segs, N, qpos, st, qproj, kproj, codes, nullptr, nullptr, nullptr, minmax, H, t0, Tl,
                                 nullptr, nullptr, rows_job(rows), zeroed);
    return rc ? rc : hept_launch_status();
}

extern "C" int hept_sort_tables_src(const float* qproj, const float* kproj, const float* eta_idx,
                                    const float* phi_idx, const float* cfac, float* minmax, int N, int H, int T,
                                    int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos, void* stream) {
    return hept_sort_tables_src_rows(qproj, kproj, eta_idx, phi_idx, cfac, minmax, N, H, T, t0, Tl, sort_ws, qpos, kpos,
                                     nullptr, stream);
}

int hept_sort_tables_src_rows(const float* qproj, const float* kproj, const float* eta_idx, const float* phi_idx,
                              const float* cfac, float* minmax, int N, int H, int T, int t0, int Tl, void* sort_ws,
                              int32_t* qpos, int32_t* kpos, const HeptRowsJob* rows, void* stream, bool zeroed) {
    if (!qproj || !kproj || !eta_idx || !phi_idx || !cfac || !minmax || !sort_ws || !qpos || !kpos)
        return HEPT_ERR_ARG;
    if (rows && (!rows->v || !rows->kvhat || rows->N != N || !hept_sort_carries_rows(N, rows->H, rows->D))) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (kpos != qpos + (size_t)Tl * H * N) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int segs = 2 * Tl * H;
    if (N <= SMALL_CAP)
        return launch_small_sort<1>(segs, st, qproj, kproj, nullptr, eta_idx, phi_idx, cfac, minmax, N, H, t0, Tl, qpos);
    const SortBuffers b = carve_sort(sort_ws, segs, N);
    hipLaunchKernelGGL(src_bound_kernel, dim3(Tl * H), dim3(SORT_THREADS), 0, st, eta_idx, phi_idx, cfac, N, H, t0,
                       minmax);
    const int rc = run_passes<1>(b, segs, N, qpos, st, qproj, kproj, nullptr, eta_idx, phi_idx, cfac, minmax, H, t0, Tl,
                                 nullptr, nullptr, rows_job(rows), zeroed);
    return rc ? rc : hept_launch_status();
}

extern "C" size_t hept_argsort_workspace_bytes(int S, int L) { return sort_bytes((size_t)S, (size_t)L); }

namespace {
int segmented_argsort_impl(const float* keys, int S, int L, const int* seg_len, void* ws, int32_t* pos, void* stream,
                           const float* bounds = nullptr) {
    if (!keys || !ws || !pos) return HEPT_ERR_ARG;
    if (S < 1 || L < 1) return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (L <= SMALL_CAP)
        return launch_small_sort<2>(S, st, keys, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, L, 1, 0, 1, pos,
                                    seg_len);
    const int n_chunks = (L + SORT_CHUNK - 1) / SORT_CHUNK;
    const SortBuffers b = carve_sort(ws, S, L);
    if (!bounds) {   // the key range of every segment: two more launches that a caller with known bounds saves
        if (hipMemsetAsync(b.range, 0xFF, (size_t)S * 8, st) != hipSuccess) return HEPT_ERR_LAUNCH;
        hipLaunchKernelGGL(raw_range_kernel, dim3(n_chunks, S), dim3(SORT_THREADS), 0, st, keys, L, seg_len, b.range);
    }
    const int rc = run_passes<2>(b, S, L, pos, st, keys, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 1, seg_len,
                                 bounds);
    return rc ? rc : hept_launch_status();
}
}  // namespace

// internal (common.h): every key lies in [lo, hi] (finite) -- the range pass is skipped
int hept_segmented_argsort_bounded(const float* keys, int S, int L, float lo, float hi, void* ws, int32_t* pos,
                                   void* stream) {
    const float bounds[2] = {lo, hi};
    return segmented_argsort_impl(keys, S, L, nullptr, ws, pos, stream, bounds);
}

extern "C" int hept_segmented_argsort(const float* keys, int S, int L, void* ws, int32_t* pos, void* stream) {
    return segmented_argsort_impl(keys, S, L, nullptr, ws, pos, stream);
}

extern "C" int hept_segmented_argsort_ragged(const float* keys, int S, int L, const int32_t* seg_len, void* ws,
                                             int32_t* pos, void* stream) {
    if (!seg_len) return HEPT_ERR_ARG;
    return segmented_argsort_impl(keys, S, L, seg_len, ws, pos, stream);
}
