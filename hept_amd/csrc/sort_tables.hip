// sort_tables: AND-shifted sort keys + exact stable segmented sort: one counting pass on the top bits of a monotone
// bucket id, then every bucket is finished inside LDS.  Three kernels.
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
//   K1 keygen       key -> order-preserving u32, per-chunk histogram of the top HEPT_TOP_BITS id bits
//   K3 scatter      counting pass on the top bits: every workgroup reduces its segment's chunk histograms to its own
//                   global offsets (no separate scan kernel), bucket-sorts its 4096-key chunk inside LDS (slot = LDS
//                   atomic on the digit counter) and writes runs of consecutive (key,index) pairs
//   K4 bucket sort  one workgroup per (segment, bucket): histogram / prefix / scatter on the LOW id bits inside LDS
//                   groups the pairs by their full id, then every pair counts the members of its id group that are
//                   smaller as u64 (key << 32 | index) -> final position.  Groups are 1-3 pairs.
// The result is the exact stable sort for ANY input; cost O(N + sum group^2).  Adversarial inputs (all keys
// inside 2^-bits of the range) degrade to O(N^2) compares per segment -- slow, never wrong.
// History (tracking-60k, 48 segments x 60 032 keys): 4-pass LSD radix 144 us -> two 8-bit LSD passes + windowed
// neighbour rank 89 us -> this design 44 us.
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

// per-segment id map parameters, written once by K1 (chunk 0) and read by the later kernels
struct SegParams {
    float kmin, scale;
};

// K1: keys0[seg][n] = ordered bits of (proj + float(code) * span); hist[seg][chunk][256] of the low id byte.
// SRC = true: the src variant's float shift (get_geo_shift, src/models/attention/hept.py:46-56) replaces
// float(code) * span:  shift = (phi * span) * cfac + eta * span, every operation rounded on its own; the
// key bound uses the third partial column as max(phi * cfac + eta).
template <bool SRC>
__global__ __launch_bounds__(SORT_THREADS) void keygen_hist_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ eta_idx, const float* __restrict__ phi_idx, const float* __restrict__ cfac,
    const float* __restrict__ minmax, int N, int H, int t0, int Tl, unsigned int* __restrict__ keys0,
    unsigned int* __restrict__ hist, SegParams* __restrict__ seg_params, int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    __shared__ float red_s[3][SORT_WAVES];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    const int th = seg % (Tl * H);  // local (table, head)
    const bool is_k = seg >= Tl * H;
    const int t = th / H, h = th % H;
    h_s[tid] = 0;

    // hash range + largest code of this (table, head): reduce the prep kernel's per-workgroup partials
    float lo = INFINITY, hi = -INFINITY, cmax = 0.f;
    {
        f32x4 m[HEPT_PREP_GRID / SORT_THREADS];
#pragma unroll
        for (int i = 0; i < HEPT_PREP_GRID / SORT_THREADS; ++i)
            m[i] = *reinterpret_cast<const f32x4*>(minmax + (((size_t)t * H + h) * HEPT_PREP_GRID + i * SORT_THREADS + tid) * 4);
#pragma unroll
        for (int i = 0; i < HEPT_PREP_GRID / SORT_THREADS; ++i) {
            lo = fminf(lo, m[i][0]);
            hi = fmaxf(hi, m[i][1]);
            cmax = fmaxf(cmax, m[i][2]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
        cmax = fmaxf(cmax, __shfl_xor(cmax, off));
    }
    if ((tid & 63) == 0) { red_s[0][tid >> 6] = lo; red_s[1][tid >> 6] = hi; red_s[2][tid >> 6] = cmax; }
    __syncthreads();
    lo = fminf(fminf(red_s[0][0], red_s[0][1]), fminf(red_s[0][2], red_s[0][3]));
    hi = fmaxf(fmaxf(red_s[1][0], red_s[1][1]), fmaxf(red_s[1][2], red_s[1][3]));
    cmax = fmaxf(fmaxf(red_s[2][0], red_s[2][1]), fmaxf(red_s[2][2], red_s[2][3]));
    const float span = hi - lo;
    // keys lie in [lo, hi + cmax*span] (codes >= 0); any key outside is clamped by bucket_id (still monotone)
    const float width = (hi + cmax * span) - lo;
    float scale = width > 0.f ? (float)ID_BUCKETS / width : 0.f;
    if (!(scale < 3.0e38f)) scale = 0.f;  // inf/nan guard for denormal widths: one bucket, still exact
    if (chunk == 0 && tid == 0) seg_params[seg] = SegParams{lo, scale};

    const float* proj = (is_k ? kproj : qproj) + (size_t)th * N;
    const size_t row_off = ((size_t)(t0 + t) * H + h) * N;
    unsigned int* kout = keys0 + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
    if constexpr (SRC) {
        const float cf = cfac[(size_t)(t0 + t) * H + h];
        const float* eta = eta_idx + row_off;
        const float* phi = phi_idx + row_off;
#pragma unroll 4
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const int n = base + i * SORT_THREADS + tid;
            if (n < N) {
                float t1 = eta[n] * span;
                asm volatile("" : "+v"(t1));
                float t2 = phi[n] * span;
                asm volatile("" : "+v"(t2));
                t2 = t2 * cf;
                asm volatile("" : "+v"(t2));
                float t3 = t2 + t1;
                asm volatile("" : "+v"(t3));
                const unsigned int u = ordered_bits(proj[n] + t3);
                kout[n] = u;
                atomicAdd(&h_s[bucket_id(u, lo, scale) >> TOP_SHIFT], 1u);
            }
        }
        __syncthreads();
        hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
        return;
    }
    const int64_t* code = codes + row_off;
    // two separately rounded ops per key, as the two eager ops of the reference; HIP's __fmul_rn /
    // __fadd_rn are plain * and + and would be contracted to one fma without -ffp-contract=off
    // (Makefile) -- the asm barrier makes it explicit here as well
    auto make_key = [&](float pj, long long cd) {
        float off = (float)cd * span;
        asm volatile("" : "+v"(off));
        return ordered_bits(pj + off);
    };
    const bool vec_ok = (N % 4) == 0;  // segment bases stay 16-B aligned
#pragma unroll
    for (int i = 0; i < SORT_ITEMS / 4; ++i) {
        const int n = base + (i * SORT_THREADS + tid) * 4;
        if (vec_ok && n + 3 < N) {
            typedef __attribute__((ext_vector_type(2))) long long i64x2;
            const f32x4 pj = *reinterpret_cast<const f32x4*>(proj + n);
            const i64x2 c01 = *reinterpret_cast<const i64x2*>(code + n);
            const i64x2 c23 = *reinterpret_cast<const i64x2*>(code + n + 2);
            u32x4 u;
            u[0] = make_key(pj[0], c01[0]);
            u[1] = make_key(pj[1], c01[1]);
            u[2] = make_key(pj[2], c23[0]);
            u[3] = make_key(pj[3], c23[1]);
            *reinterpret_cast<u32x4*>(kout + n) = u;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&h_s[bucket_id(u[e], lo, scale) >> TOP_SHIFT], 1u);
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < N) {
                    const unsigned int u = make_key(proj[n + e], code[n + e]);
                    kout[n + e] = u;
                    atomicAdd(&h_s[bucket_id(u, lo, scale) >> TOP_SHIFT], 1u);
                }
        }
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

// K3: counting-sort pass on the top id bits: keys0 (index implicit) -> (key, index) pairs grouped into the NTOP
// buckets of their segment (any order inside a bucket: K4 ranks full pairs).  A chunk of 4096 keys is bucket-sorted
// inside LDS first (local slot = LDS atomic on the digit counter), so that the global writes are runs of
// consecutive pairs instead of single 8-byte scatters.
// EMBED (segments of at most 2^20 keys): the pair carries the low id bits next to the point index,
//     pair = key << 32 | low id << 20 | index,
// so that K4 reads a pair's group with a shift instead of recomputing the float id map (K4 is bound by VALU issue:
// three id evaluations per pair were 40 % of its instructions).  The order of the pairs as u64 is unchanged: the id is
// a monotone function of the key, equal keys carry equal ids.
constexpr int EMBED_SHIFT = 20;
constexpr unsigned int EMBED_INDEX_MASK = (1u << EMBED_SHIFT) - 1u;
static_assert(TOP_SHIFT + EMBED_SHIFT <= 32, "low id bits + index fit the low word of a pair");
// SCT threads per workgroup (the chunk stays 4096 keys): the launch is a single round of workgroups (720 at
// tracking-60k on 1024 resident slots), so its length is one workgroup's dependent chain; 512 threads halve the
// per-thread item loops of that chain and double the waves that hide its latencies.  The RADIX digit counters are
// owned by the first RADIX threads.
#ifndef HEPT_SCATTER_THREADS
#define HEPT_SCATTER_THREADS 512
#endif
constexpr int SCT = HEPT_SCATTER_THREADS;
constexpr int SCT_ITEMS = SORT_CHUNK / SCT;
static_assert(SCT >= RADIX && SCT % RADIX == 0 && SORT_CHUNK % SCT == 0, "digit ownership / items per thread");
template <bool EMBED>
__global__ __launch_bounds__(SCT) void scatter_kernel(const unsigned int* __restrict__ keys0,
                                                      const SegParams* __restrict__ seg_params,
                                                      const unsigned int* __restrict__ hist, int N, int n_chunks,
                                                      unsigned int* __restrict__ bstart,
                                                      unsigned long long* __restrict__ dst_pairs,
                                                      const int* __restrict__ seg_len) {
    __shared__ unsigned long long stage_s[SORT_CHUNK];   // the chunk, digit-sorted (32 KiB)
    __shared__ unsigned int cnt_s[RADIX];                // keys of the chunk per digit
    __shared__ unsigned int start_s[RADIX];              // first local position of a digit
    __shared__ unsigned int goff_s[RADIX];               // global offset of the digit's first key of this chunk
    __shared__ unsigned int wsum_s[RADIX / HEPT_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool digit = tid < RADIX;                      // this thread owns digit `tid`
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const SegParams rg = seg_params[seg];
    const size_t seg_off = (size_t)seg * N;        // N = segment stride; len = keys that take part (ragged argsort)
    const int len = seg_len ? seg_len[seg] : N;
    if (digit) cnt_s[tid] = 0;
    unsigned int key[SCT_ITEMS];
    const int base = chunk * SORT_CHUNK;
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int n = base + r * SCT + tid;
        key[r] = n < len ? keys0[seg_off + n] : 0u;
    }
    // global offset of digit `tid` for this chunk = (keys of the segment with a smaller digit) + (same digit in
    // earlier chunks): every workgroup reduces the segment's chunk histograms itself (a few KiB from L2) instead
    // of waiting for a separate scan kernel
    unsigned int own = 0, tot = 0;
    if (digit) {
        const unsigned int* hseg = hist + (size_t)seg * n_chunks * RADIX + tid;
        for (int c0 = 0; c0 < n_chunks; c0 += 8) {
            unsigned int x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (c0 + i < n_chunks) ? hseg[(size_t)(c0 + i) * RADIX] : 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                tot += x[i];
                own += (c0 + i < chunk) ? x[i] : 0u;
            }
        }
    }
    {
        unsigned int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (digit && lane == 63) wsum_s[w] = incl;
        __syncthreads();
        if (digit) {
            unsigned int excl = incl - tot;
#pragma unroll
            for (int ww = 0; ww < RADIX / HEPT_WAVE; ++ww)
                if (ww < w) excl += wsum_s[ww];
            goff_s[tid] = excl + own;
            if (chunk == 0) bstart[(size_t)seg * RADIX + tid] = excl;
        }
    }
    __syncthreads();
    unsigned short rank[SCT_ITEMS], lowid[SCT_ITEMS];
    unsigned char dig[SCT_ITEMS];
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int n = base + r * SCT + tid;
        const unsigned int id = bucket_id(key[r], rg.kmin, rg.scale);
        const unsigned int dg = id >> TOP_SHIFT;
        lowid[r] = (unsigned short)(id & (unsigned int)(LOBINS - 1));
        dig[r] = (unsigned char)dg;
        rank[r] = n < len ? (unsigned short)atomicAdd(&cnt_s[dg], 1u) : (unsigned short)0;
    }
    __syncthreads();
    // digit `tid`: exclusive prefix over the digits -> first local position of the digit
    const unsigned int total = digit ? cnt_s[tid] : 0u;
    unsigned int incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (digit && lane == 63) wsum_s[w] = incl;
    __syncthreads();
    if (digit) {
        unsigned int first = incl - total;
#pragma unroll
        for (int ww = 0; ww < RADIX / HEPT_WAVE; ++ww)
            if (ww < w) first += wsum_s[ww];
        start_s[tid] = first;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int n = base + r * SCT + tid;
        if (n < len)
            stage_s[start_s[dig[r]] + rank[r]] = ((unsigned long long)key[r] << 32) |
                                                 (EMBED ? ((unsigned int)lowid[r] << EMBED_SHIFT) | (unsigned int)n : (unsigned int)n);
    }
    __syncthreads();
    // write out: consecutive local positions of one digit are consecutive global positions
    const int n_valid = min(SORT_CHUNK, len - base);
#pragma unroll
    for (int r = 0; r < SCT_ITEMS; ++r) {
        const int lp = r * SCT + tid;
        if (lp < n_valid) {
            const unsigned long long p = stage_s[lp];
            const unsigned int dg = bucket_id((unsigned int)(p >> 32), rg.kmin, rg.scale) >> TOP_SHIFT;
            dst_pairs[seg_off + goff_s[dg] + (lp - start_s[dg])] = p;
        }
    }
}

// K4: every (segment, top-bits bucket) is finished by one workgroup.  A bucket holds the pairs whose id shares the
// top bits, contiguous after K3; id is monotone in the key, so the bucket's final positions are exactly its own
// range [start, end) and only the order inside it is left.  Histogram of the LOW id bits -> exclusive prefix
// -> scatter (any order): the pairs are now grouped by their full id, group g = [first[g], first[g+1]), and
//        final position(i) = start + first[g] + #{ j in group g : pair_j < pair_i }        (pairs are unique u64)
// Groups are 1-3 pairs at tracking-60k.  A bucket larger than the LDS tile takes the same three steps through a
// global scratch copy, grouped by sampled splitters instead of id bits (streaming; slower, never wrong); a
// tile-sized bucket of equal keys costs at most CAP^2 compares.
// The whole bucket is loaded into registers with every load in flight at once; only the grouped copy lives in LDS.
// One bins array serves as histogram, exclusive prefix and scatter cursor: after the scatter cur[d] is one past the
// last slot of group d, i.e. group d = [cur[d-1], cur[d]).
constexpr int BKT_THREADS = HEPT_BKT_THREADS;
constexpr int BKT_WAVES = BKT_THREADS / HEPT_WAVE;
constexpr int BKT_BINS_PER_THREAD = LOBINS / BKT_THREADS;
static_assert(LOBINS % BKT_THREADS == 0, "every thread owns the same number of bins");
template <int CAP, bool EMBED>
__global__ __launch_bounds__(BKT_THREADS) void bucket_sort_kernel(const unsigned long long* __restrict__ pairs,
                                                                  unsigned long long* __restrict__ scratch,
                                                                  const SegParams* __restrict__ seg_params,
                                                                  const unsigned int* __restrict__ bstart, int N,
                                                                  int* __restrict__ pos_out,
                                                                  const int* __restrict__ seg_len) {
    static_assert(CAP >= LOBINS, "the tile also holds the splitters of the streaming path");
    __shared__ unsigned long long tile_s[CAP];
    __shared__ unsigned int cur_s[LOBINS + 1];  // [0] stays 0; bin d lives at [d + 1]
    __shared__ unsigned int wsum_s[BKT_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.y, bucket = blockIdx.x;
    const unsigned int* o = bstart + (size_t)seg * RADIX;  // keys of the segment with a smaller digit (written by K3)
    const int start = (int)o[bucket];
    const int end = bucket == NTOP - 1 ? (seg_len ? seg_len[seg] : N) : (int)o[bucket + 1];
    const int nb = end - start;
    if (nb <= 0) return;
    const unsigned long long* src = pairs + (size_t)seg * N + start;
    int* out = pos_out + (size_t)seg * N + start;
    const SegParams rg = seg_params[seg];
    // Grouping key inside the bucket.  LDS path: the low bits of the global id; a bucket of up to 2 CAP pairs (skewed
    // key distributions -- pileup clouds of unequal size -- put 1.5x to 2x the expected pairs into a few buckets) goes
    // through the tile in two passes, lower half of the id bins first, if each half fits.  Streaming path (anything
    // else: a pile of equal or nearly equal keys, possibly next to a few spread ones, which neither those bits nor
    // any linear map of the key range can separate): SPLITTERS -- LOBINS pairs sampled at regular positions of the
    // bucket and sorted in LDS; a pair's bin is the number of splitters <= it (binary search).  Pairs are unique
    // u64, so even 60 000 equal keys spread over the bins by their index.
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
    unsigned int* bin_s = cur_s + 1;
    auto zero_bins = [&]() {
#pragma unroll
        for (int u = 0; u < BKT_BINS_PER_THREAD; ++u) bin_s[u * BKT_THREADS + tid] = 0;
        if (tid == 0) cur_s[0] = 0;
    };
    auto prefix_bins = [&]() {  // exclusive prefix over the bins: thread owns bins BPT*tid .. BPT*tid + BPT - 1
        unsigned int c[BKT_BINS_PER_THREAD], tot = 0;
#pragma unroll
        for (int u = 0; u < BKT_BINS_PER_THREAD; ++u) { c[u] = bin_s[BKT_BINS_PER_THREAD * tid + u]; tot += c[u]; }
        unsigned int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
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
    zero_bins();
    // The first CAP pairs of the bucket are loaded into registers with every load in flight at once; only the
    // grouped copy lives in LDS.  One bins array serves as histogram, exclusive prefix and scatter cursor: after
    // the scatter cur[d] is one past the last slot of group d, i.e. group d = [cur[d-1], cur[d]).
    constexpr int ITEMS = CAP / BKT_THREADS;
    unsigned long long mine[ITEMS];
    if (in_lds) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int i = u * BKT_THREADS + tid;
            mine[u] = i < nb ? src[i] : 0ull;
        }
    }
    __syncthreads();
    int n_low = nb;  // pairs in the lower half of the id bins (two-pass buckets)
    if (in_lds) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u)
            if (u * BKT_THREADS + tid < nb) atomicAdd(&bin_s[lo_of(mine[u])], 1u);
        for (int i = CAP + tid; i < nb; i += BKT_THREADS) atomicAdd(&bin_s[lo_of(src[i])], 1u);
        __syncthreads();
        prefix_bins();
        if (nb > CAP) {
            n_low = (int)bin_s[LOBINS / 2];
            if (n_low > CAP || nb - n_low > CAP) in_lds = false;  // uniform: every thread reads the same word
        }
    }
    if (!in_lds) {
        __syncthreads();
        for (int i = tid; i < LOBINS; i += BKT_THREADS) spl_s[i] = src[(size_t)i * nb / LOBINS];
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
        for (int i = tid; i < nb; i += BKT_THREADS) atomicAdd(&bin_s[lo_of(src[i])], 1u);
        __syncthreads();
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
            for (int u = 0; u < ITEMS; ++u)
                if (u * BKT_THREADS + tid < nb) {
                    const unsigned int d = lo_of(mine[u]);
                    if (n_pass == 1 || (d >> (TOP_SHIFT - 1)) == half) tile_s[atomicAdd(&bin_s[d], 1u) - off] = mine[u];
                }
            for (int i = CAP + tid; i < nb; i += BKT_THREADS) {
                const unsigned long long p = src[i];
                const unsigned int d = lo_of(p);
                if ((d >> (TOP_SHIFT - 1)) == half) tile_s[atomicAdd(&bin_s[d], 1u) - off] = p;
            }
            __syncthreads();
            rank_all(tile_s, off, cnt);
            __syncthreads();
        }
    } else {
        unsigned long long* g = scratch + (size_t)seg * N + start;
        for (int i = tid; i < nb; i += BKT_THREADS) {
            const unsigned long long p = src[i];
            g[atomicAdd(&bin_s[lo_of(p)], 1u)] = p;
        }
        __threadfence_block();
        __syncthreads();
        rank_all(g, 0, nb);
    }
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

// keys0 = ordered bits of the raw keys + per-chunk histogram of the low id byte
__global__ __launch_bounds__(SORT_THREADS) void raw_keygen_hist_kernel(const float* __restrict__ keys, int L,
                                                                       const int* __restrict__ seg_len,
                                                                       const unsigned int* __restrict__ range_bits,
                                                                       SegParams* __restrict__ seg_params,
                                                                       unsigned int* __restrict__ keys0,
                                                                       unsigned int* __restrict__ hist, int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    SegParams rg;
    {
        const unsigned int lo_b = range_bits[2 * seg], nhi_b = range_bits[2 * seg + 1];
        const bool none = lo_b == 0xFFFFFFFFu && nhi_b == 0xFFFFFFFFu;  // no finite key in the segment
        const float lo = none ? 0.f : from_ordered(lo_b), hi = none ? 0.f : from_ordered(~nhi_b);
        const float width = hi - lo;
        float scale = width > 0.f ? (float)ID_BUCKETS / width : 0.f;
        if (!(scale < 3.0e38f)) scale = 0.f;
        rg = SegParams{lo, scale};
        if (chunk == 0 && tid == 0) seg_params[seg] = rg;
    }
    h_s[tid] = 0;
    __syncthreads();
    const int base = chunk * SORT_CHUNK;
    const int len = seg_len ? seg_len[seg] : L;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < len) {
            const unsigned int u = ordered_bits(keys[(size_t)seg * L + n]);
            keys0[(size_t)seg * L + n] = u;
            atomicAdd(&h_s[bucket_id(u, rg.kmin, rg.scale) >> TOP_SHIFT], 1u);
        }
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
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

// MODE 0: key = proj + float(code) * span;  MODE 1: src variant (get_geo_shift);  MODE 2: raw keys (S segments of L)
template <int MODE>
__global__ __launch_bounds__(SMALL_THREADS) void small_sort_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ eta_idx, const float* __restrict__ phi_idx, const float* __restrict__ cfac,
    const float* __restrict__ minmax, int N_stride, int H, int t0, int Tl, int* __restrict__ pos_out,
    const int* __restrict__ seg_len) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long small_tile_s[];  // [SMALL_CAP] pairs, then bins
    __shared__ float red_s[3][SMALL_WAVES];
    __shared__ unsigned int wsum_s[SMALL_WAVES];
    unsigned int* cur_s = reinterpret_cast<unsigned int*>(small_tile_s + SMALL_CAP);  // [0] stays 0; bin d at [d + 1]
    unsigned int* bin_s = cur_s + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, seg = blockIdx.x;
    const int N = seg_len ? seg_len[seg] : N_stride;  // keys that take part (ragged argsort); N_stride = segment pitch
    for (int i = tid; i < SMALL_BINS + 1; i += SMALL_THREADS) cur_s[i] = 0;

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
    float cf = 0.f;
    if (MODE == 1) { const int th = seg % (Tl * H); cf = cfac[(size_t)(t0 + th / H) * H + th % H]; }
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u) {
        const int n = u * SMALL_THREADS + tid;
        mine[u] = ~0ull;
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
            mine[u] = ((unsigned long long)ub << 32) | (unsigned int)n;
            atomicAdd(&bin_s[bin_of(ub)], 1u);
        }
    }
    __syncthreads();
    {   // exclusive prefix over the bins: thread owns 4 consecutive bins
        constexpr int BPT = SMALL_BINS / SMALL_THREADS;
        unsigned int c[BPT], tot = 0;
#pragma unroll
        for (int u = 0; u < BPT; ++u) { c[u] = bin_s[BPT * tid + u]; tot += c[u]; }
        unsigned int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wsum_s[w] = incl;
        __syncthreads();
        unsigned int run = incl - tot;
        for (int ww = 0; ww < w; ++ww) run += wsum_s[ww];
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            bin_s[BPT * tid + u] = run;
            run += c[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SMALL_ITEMS; ++u)
        if (u * SMALL_THREADS + tid < N)
            small_tile_s[atomicAdd(&bin_s[bin_of((unsigned int)(mine[u] >> 32))], 1u)] = mine[u];
    __syncthreads();
    int* out = pos_out + (size_t)seg * N_stride;
    for (int i = tid; i < N; i += SMALL_THREADS) {
        const unsigned long long p = small_tile_s[i];
        const unsigned int d = bin_of((unsigned int)(p >> 32));
        const int g0 = (int)cur_s[d], g1 = (int)cur_s[d + 1];
        int smaller = 0;
        for (int j = g0; j < g1; j += 4) {
            unsigned long long q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = small_tile_s[min(j + u, g1 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) smaller += (j + u < g1) && (q[u] < p);
        }
        out[g0 + smaller] = (int)(unsigned int)p;
    }
}

template <int MODE>
int launch_small_sort(int segs, hipStream_t st, const float* qproj, const float* kproj, const int64_t* codes,
                      const float* eta, const float* phi, const float* cfac, const float* minmax, int N, int H, int t0,
                      int Tl, int* pos, const int* seg_len = nullptr) {
    static LdsRaised raised;
    if (hept_raise_lds(raised, reinterpret_cast<const void*>(small_sort_kernel<MODE>), SMALL_LDS)) return HEPT_ERR_LAUNCH;
    hipLaunchKernelGGL(small_sort_kernel<MODE>, dim3(segs), dim3(SMALL_THREADS), SMALL_LDS, st, qproj, kproj, codes, eta,
                       phi, cfac, minmax, N, H, t0, Tl, pos, seg_len);
    return hept_launch_status();
}

// the passes shared by hept_sort_tables and hept_segmented_argsort: keys0 + hist(top bits) + params -> pos
struct SortBuffers {
    unsigned int* keys0;
    unsigned long long *pa, *pb;
    unsigned int* hist;
    unsigned int* bstart;  // [segs][RADIX] first position of every top-level bucket
    unsigned int* range;   // [segs][2] ordered-uint finite min / ~max of raw keys (hept_segmented_argsort)
    SegParams* params;
};
SortBuffers carve_sort(void* sort_ws, int segs, int N) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    char* ws = reinterpret_cast<char*>(sort_ws);
    SortBuffers b;
    b.keys0 = reinterpret_cast<unsigned int*>(ws);
    ws += al((size_t)segs * N * 4);
    b.pa = reinterpret_cast<unsigned long long*>(ws);
    ws += al((size_t)segs * N * 8);
    b.pb = reinterpret_cast<unsigned long long*>(ws);
    ws += al((size_t)segs * N * 8);
    b.hist = reinterpret_cast<unsigned int*>(ws);
    ws += al((size_t)segs * n_chunks * RADIX * 4);
    b.bstart = reinterpret_cast<unsigned int*>(ws);
    ws += al((size_t)segs * RADIX * 4);
    b.range = reinterpret_cast<unsigned int*>(ws);
    ws += al((size_t)segs * 8);
    b.params = reinterpret_cast<SegParams*>(ws);
    return b;
}
constexpr int BKT_CAP_SMALL = HEPT_BKT_CAP;  // LDS tile: the average bucket is N/NTOP
constexpr int BKT_CAP_LARGE = 6 * HEPT_BKT_CAP;  // 48 KiB tile for longer segments (average bucket up to ~3000 pairs)
template <bool EMBED>
void run_passes_impl(const SortBuffers& b, int segs, int N, int* pos, hipStream_t st, const int* seg_len) {
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const dim3 grid(n_chunks, segs), block(SCT);
    hipLaunchKernelGGL(scatter_kernel<EMBED>, grid, block, 0, st, b.keys0, b.params, b.hist, N, n_chunks, b.bstart, b.pa,
                       seg_len);
    const dim3 grid4(NTOP, segs);
    if ((size_t)N <= (size_t)NTOP * (BKT_CAP_SMALL / 2))
        hipLaunchKernelGGL((bucket_sort_kernel<BKT_CAP_SMALL, EMBED>), grid4, dim3(BKT_THREADS), 0, st, b.pa, b.pb, b.params,
                           b.bstart, N, pos, seg_len);
    else
        hipLaunchKernelGGL((bucket_sort_kernel<BKT_CAP_LARGE, EMBED>), grid4, dim3(BKT_THREADS), 0, st, b.pa, b.pb, b.params,
                           b.bstart, N, pos, seg_len);
}
void run_passes(const SortBuffers& b, int segs, int N, int* pos, hipStream_t st, const int* seg_len = nullptr) {
    if (N <= (1 << EMBED_SHIFT)) run_passes_impl<true>(b, segs, N, pos, st, seg_len);
    else run_passes_impl<false>(b, segs, N, pos, st, seg_len);
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_bytes(size_t segs, size_t N) {
    const size_t n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    return align256(segs * N * 4) + 2 * align256(segs * N * 8) + align256(segs * n_chunks * RADIX * 4) +
           align256(segs * RADIX * 4) + align256(segs * 8) + align256(segs * sizeof(SegParams));
}

extern "C" size_t hept_sort_workspace_bytes(int N, int H, int Tl) { return sort_bytes((size_t)2 * Tl * H, N); }

extern "C" int hept_sort_tables(const float* qproj, const float* kproj, const int64_t* codes, const float* minmax,
                                int N, int H, int T, int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos,
                                void* stream) {
    if (!qproj || !kproj || !codes || !minmax || !sort_ws || !qpos || !kpos) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (kpos != qpos + (size_t)Tl * H * N) return HEPT_ERR_ARG;  // one (2,Tl,H,N) array: q then k
    hipStream_t st = (hipStream_t)stream;
    const int segs = 2 * Tl * H;
    if (N <= SMALL_CAP)
        return launch_small_sort<0>(segs, st, qproj, kproj, codes, nullptr, nullptr, nullptr, minmax, N, H, t0, Tl, qpos);
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const SortBuffers b = carve_sort(sort_ws, segs, N);
    hipLaunchKernelGGL(keygen_hist_kernel<false>, dim3(n_chunks, segs), dim3(SORT_THREADS), 0, st, qproj, kproj, codes,
                       nullptr, nullptr, nullptr, minmax, N, H, t0, Tl, b.keys0, b.hist, b.params, n_chunks);
    run_passes(b, segs, N, qpos, st);
    return hept_launch_status();
}

extern "C" int hept_sort_tables_src(const float* qproj, const float* kproj, const float* eta_idx,
                                    const float* phi_idx, const float* cfac, float* minmax, int N, int H, int T,
                                    int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos, void* stream) {
    if (!qproj || !kproj || !eta_idx || !phi_idx || !cfac || !minmax || !sort_ws || !qpos || !kpos)
        return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (kpos != qpos + (size_t)Tl * H * N) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int segs = 2 * Tl * H;
    if (N <= SMALL_CAP)
        return launch_small_sort<1>(segs, st, qproj, kproj, nullptr, eta_idx, phi_idx, cfac, minmax, N, H, t0, Tl, qpos);
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const SortBuffers b = carve_sort(sort_ws, segs, N);
    hipLaunchKernelGGL(src_bound_kernel, dim3(Tl * H), dim3(SORT_THREADS), 0, st, eta_idx, phi_idx, cfac, N, H, t0,
                       minmax);
    hipLaunchKernelGGL(keygen_hist_kernel<true>, dim3(n_chunks, segs), dim3(SORT_THREADS), 0, st, qproj, kproj, nullptr,
                       eta_idx, phi_idx, cfac, minmax, N, H, t0, Tl, b.keys0, b.hist, b.params, n_chunks);
    run_passes(b, segs, N, qpos, st);
    return hept_launch_status();
}

extern "C" size_t hept_argsort_workspace_bytes(int S, int L) { return sort_bytes((size_t)S, (size_t)L); }

namespace {
int segmented_argsort_impl(const float* keys, int S, int L, const int* seg_len, void* ws, int32_t* pos, void* stream) {
    if (!keys || !ws || !pos) return HEPT_ERR_ARG;
    if (S < 1 || L < 1) return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (L <= SMALL_CAP)
        return launch_small_sort<2>(S, st, keys, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, L, 1, 0, 1, pos,
                                    seg_len);
    const int n_chunks = (L + SORT_CHUNK - 1) / SORT_CHUNK;
    const SortBuffers b = carve_sort(ws, S, L);
    if (hipMemsetAsync(b.range, 0xFF, (size_t)S * 8, st) != hipSuccess) return HEPT_ERR_LAUNCH;
    hipLaunchKernelGGL(raw_range_kernel, dim3(n_chunks, S), dim3(SORT_THREADS), 0, st, keys, L, seg_len, b.range);
    hipLaunchKernelGGL(raw_keygen_hist_kernel, dim3(n_chunks, S), dim3(SORT_THREADS), 0, st, keys, L, seg_len, b.range,
                       b.params, b.keys0, b.hist, n_chunks);
    run_passes(b, S, L, pos, st, seg_len);
    return hept_launch_status();
}
}  // namespace

extern "C" int hept_segmented_argsort(const float* keys, int S, int L, void* ws, int32_t* pos, void* stream) {
    return segmented_argsort_impl(keys, S, L, nullptr, ws, pos, stream);
}

extern "C" int hept_segmented_argsort_ragged(const float* keys, int S, int L, const int32_t* seg_len, void* ws,
                                             int32_t* pos, void* stream) {
    if (!seg_len) return HEPT_ERR_ARG;
    return segmented_argsort_impl(keys, S, L, seg_len, ws, pos, stream);
}
