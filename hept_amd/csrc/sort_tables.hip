// sort_tables: AND-shifted sort keys + stable segmented sort (two 8-bit passes on a 16-bit bucket id + in-bucket ranking).
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
// Integer/byte work, HBM/L2-bound.  Every key gets a MONOTONE 16-bit bucket id
//     id(key) = trunc((key - kmin) * 65536 / (kmax - kmin)),
// where [kmin,kmax] = [hash min, hash max + largest code * span] is known before any key exists (from the
// prep kernel's partials).  id() is a monotone non-decreasing function of the key itself (fp32 subtract,
// multiply and truncate are monotone), so id order never contradicts key order and equal keys share an id.
//   K1 keygen      key -> order-preserving u32, per-chunk histogram of the LOW id byte
//   K2 scan        histogram -> exclusive offsets [segment][chunk][256]
//   K3 scatter     stable counting-sort pass on the low byte (wave-level ballots, 64-wide; no data-path
//                  atomics); the chunk is bucket-sorted in LDS first so that the global writes are runs of
//                  ~16 consecutive (key,index) pairs instead of single 8-byte scatters
//   K4 hist        per-chunk histogram of the HIGH id byte of the pass-1 order;  K5 = K2;  K6 = K3 (high byte)
//   K7 rank        keys are now grouped by id (~N/65536 * clumping, a handful per id): every element counts the
//                  same-id neighbours that sort before it (u64 compare of (key << 32 | index): inside an id the
//                  stable passes kept ascending index order) -> final position
// The result is the exact stable sort for ANY input; cost O(N + sum group^2).  Adversarial inputs (all keys
// inside 1/65536 of the range) degrade to O(N^2) neighbour scans per segment — slow, never wrong.
#include "common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / HEPT_WAVE;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int RADIX = 256;
constexpr int ID_BUCKETS = 65536;

__device__ __forceinline__ unsigned int ordered_bits(float key) {
    if (key == 0.f) key = 0.f;  // -0.0 and +0.0 compare equal in the reference sort
    const unsigned int u = __float_as_uint(key);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned int u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// monotone 16-bit bucket id; scale = 65536 / (kmax - kmin), 0 when all keys are equal
__device__ __forceinline__ unsigned int id16_of(unsigned int u, float kmin, float scale) {
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
    // keys lie in [lo, hi + cmax*span] (codes >= 0); any key outside is clamped by id16_of (still monotone)
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
                atomicAdd(&h_s[id16_of(u, lo, scale) & 0xFF], 1u);
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
            for (int e = 0; e < 4; ++e) atomicAdd(&h_s[id16_of(u[e], lo, scale) & 0xFF], 1u);
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < N) {
                    const unsigned int u = make_key(proj[n + e], code[n + e]);
                    kout[n + e] = u;
                    atomicAdd(&h_s[id16_of(u, lo, scale) & 0xFF], 1u);
                }
        }
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

// K4: histogram of the high id byte over the pass-1 order
__global__ __launch_bounds__(SORT_THREADS) void hist_hi_kernel(const unsigned long long* __restrict__ pairs,
                                                               const SegParams* __restrict__ seg_params, int N,
                                                               unsigned int* __restrict__ hist, int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    const SegParams rg = seg_params[seg];
    h_s[tid] = 0;
    __syncthreads();
    const uint2* src = reinterpret_cast<const uint2*>(pairs + (size_t)seg * N);
    const int base = chunk * SORT_CHUNK;
#pragma unroll 8
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < N) atomicAdd(&h_s[id16_of(src[n].y, rg.kmin, rg.scale) >> 8], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

// K2/K5: one workgroup per segment, one thread per digit.
// hist[seg][c][d] <- (keys of this segment with a smaller digit) + (same digit in earlier chunks)
__global__ __launch_bounds__(RADIX) void scan_kernel(unsigned int* __restrict__ hist, int n_chunks) {
    constexpr int BATCH = 16;
    __shared__ unsigned int wsum_s[RADIX / HEPT_WAVE];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6, seg = blockIdx.x;
    unsigned int* hseg = hist + (size_t)seg * n_chunks * RADIX + d;
    unsigned int total = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += BATCH) {
        unsigned int x[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) x[i] = (c0 + i < n_chunks) ? hseg[(size_t)(c0 + i) * RADIX] : 0u;
#pragma unroll
        for (int i = 0; i < BATCH; ++i) total += x[i];
    }
    unsigned int incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum_s[w] = incl;
    __syncthreads();
    unsigned int run = incl - total;
    for (int ww = 0; ww < w; ++ww) run += wsum_s[ww];
    for (int c0 = 0; c0 < n_chunks; c0 += BATCH) {
        unsigned int x[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) x[i] = (c0 + i < n_chunks) ? hseg[(size_t)(c0 + i) * RADIX] : 0u;
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (c0 + i < n_chunks) hseg[(size_t)(c0 + i) * RADIX] = run;
            run += x[i];
        }
    }
}

// K3/K6: one stable counting-sort pass on one id byte.  HI = false: source is keys0 (index implicit), digit =
// low byte; HI = true: source is the pass-1 pairs, digit = high byte.
template <bool HI>
__global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(const unsigned int* __restrict__ keys0,
                                                               const unsigned long long* __restrict__ src_pairs,
                                                               const SegParams* __restrict__ seg_params,
                                                               const unsigned int* __restrict__ offs, int N,
                                                               int n_chunks,
                                                               unsigned long long* __restrict__ dst_pairs) {
    __shared__ unsigned long long stage_s[SORT_CHUNK];   // the chunk, digit-sorted (32 KiB)
    __shared__ unsigned int cnt_s[SORT_WAVES][RADIX];    // per-wave digit counters -> exclusive wave prefix
    __shared__ unsigned int start_s[RADIX];              // first local position of a digit
    __shared__ unsigned int goff_s[RADIX];               // global offset of the digit's first key of this chunk
    __shared__ unsigned int wsum_s[SORT_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const SegParams rg = seg_params[seg];
    const size_t seg_off = (size_t)seg * N;
#pragma unroll
    for (int ww = 0; ww < SORT_WAVES; ++ww) cnt_s[ww][tid] = 0;
    goff_s[tid] = offs[((size_t)seg * n_chunks + chunk) * RADIX + tid];

    // wave w owns 1024 consecutive keys, 16 rounds of 64 (stable: index order)
    unsigned long long pr[SORT_ITEMS];
    const int wbase = chunk * SORT_CHUNK + w * (SORT_ITEMS * HEPT_WAVE);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        if (HI)
            pr[r] = n < N ? src_pairs[seg_off + n] : ~0ull;
        else
            pr[r] = n < N ? (((unsigned long long)keys0[seg_off + n] << 32) | (unsigned int)n) : ~0ull;
    }
    __syncthreads();
    unsigned short rank[SORT_ITEMS];
    unsigned char dig[SORT_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        const bool valid = n < N;
        const unsigned int id = id16_of((unsigned int)(pr[r] >> 32), rg.kmin, rg.scale);
        const unsigned int dg = valid ? (HI ? id >> 8 : id & 0xFF) : 0xFFu;
        dig[r] = (unsigned char)dg;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const unsigned int prior = cnt_s[w][dg];
        const unsigned int ahead = __popcll(peers & lt_mask);
        if (valid && ahead == 0) cnt_s[w][dg] = prior + __popcll(peers);
        rank[r] = (unsigned short)(prior + ahead);
    }
    __syncthreads();
    // digit `tid`: exclusive prefix over the waves, then over the digits -> first local position of the digit
    unsigned int total = 0;
#pragma unroll
    for (int ww = 0; ww < SORT_WAVES; ++ww) {
        const unsigned int c = cnt_s[ww][tid];
        cnt_s[ww][tid] = total;
        total += c;
    }
    unsigned int incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum_s[w] = incl;
    __syncthreads();
    unsigned int first = incl - total;
#pragma unroll
    for (int ww = 0; ww < SORT_WAVES; ++ww)
        if (ww < w) first += wsum_s[ww];
    start_s[tid] = first;
    __syncthreads();
    // bucket-sort the chunk inside LDS
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        if (n < N) stage_s[start_s[dig[r]] + cnt_s[w][dig[r]] + rank[r]] = pr[r];
    }
    __syncthreads();
    // write out: consecutive local positions of one digit are consecutive global positions
    const int n_valid = min(SORT_CHUNK, N - chunk * SORT_CHUNK);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int lp = r * SORT_THREADS + tid;
        if (lp < n_valid) {
            const unsigned long long p = stage_s[lp];
            const unsigned int id = id16_of((unsigned int)(p >> 32), rg.kmin, rg.scale);
            const unsigned int dg = HI ? id >> 8 : id & 0xFF;
            dst_pairs[seg_off + goff_s[dg] + (lp - start_s[dg])] = p;
        }
    }
}

// K7: final position of every element.  After the two passes the pairs are ordered by id, and id is monotone
// in the key, so everything left of an element's id group is smaller and everything right of it is larger as
// u64 (key << 32 | index); inside the group the stable passes kept ascending index.  Hence, for ANY window
// [i-K, i+K] that contains the whole group:
//        final position(i) = (i - K) + #{ j in window : pair_j < pair_i }
// -- a branch-free count with no id logic (slots left of the segment hold 0, slots right of it ~0, which
// makes the formula exact at the segment ends as well).  A thread owns 4 consecutive positions and streams
// the staged pairs of their joint window once.  Only when a group reaches a window edge (groups are ~5
// keys at tracking-60k; adversarial inputs can make them arbitrarily long) the element is recounted with
// plain loops over its whole group.
constexpr int RANK_PER_THREAD = 4;
constexpr int RANK_SPAN = SORT_THREADS * RANK_PER_THREAD;  // positions per workgroup
constexpr int RANK_K = 24;
__global__ __launch_bounds__(SORT_THREADS) void neighbour_rank_kernel(const unsigned long long* __restrict__ pairs,
                                                                      const SegParams* __restrict__ seg_params,
                                                                      int N, int* __restrict__ pos_out) {
    constexpr int W = RANK_SPAN + 2 * RANK_K;
    // slot(g) = g + g/4: a thread's window starts at a multiple of 4, so lane t reads slot 5t + const ->
    // 10-dword lane stride, bank-conflict free for ds_read_b64 (a plain layout is 4-way conflicted)
    __shared__ unsigned long long p_s[W + W / 4 + 1];
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int i0 = blockIdx.x * RANK_SPAN;
    const SegParams rg = seg_params[seg];
    const unsigned long long* pr = pairs + (size_t)seg * N;
    auto id_of = [&](unsigned long long p) { return id16_of((unsigned int)(p >> 32), rg.kmin, rg.scale); };
    for (int j = tid; j < W; j += SORT_THREADS) {
        const int g = i0 - RANK_K + j;
        p_s[j + (j >> 2)] = g < 0 ? 0ull : (g < N ? pr[g] : ~0ull);
    }
    __syncthreads();
    const int b = i0 + RANK_PER_THREAD * tid;  // first of this thread's 4 positions
    if (b >= N) return;
    const unsigned long long* win0 = p_s + 5 * tid;  // window slot j of this thread lives at win0[j + j/4]
#define WIN(j) win0[(j) + ((j) >> 2)]
    unsigned long long mine[RANK_PER_THREAD];
    int cnt[RANK_PER_THREAD];
#pragma unroll
    for (int e = 0; e < RANK_PER_THREAD; ++e) {
        mine[e] = WIN(RANK_K + e);
        cnt[e] = 0;
    }
#pragma unroll
    for (int j = 0; j < 2 * RANK_K + RANK_PER_THREAD; ++j) {
        const unsigned long long p = WIN(j);
#pragma unroll
        for (int e = 0; e < RANK_PER_THREAD; ++e)
            if (j >= e && j <= e + 2 * RANK_K) cnt[e] += p < mine[e];  // window of element e: [e, e + 2K]
    }
#pragma unroll
    for (int e = 0; e < RANK_PER_THREAD; ++e) {
        const int i = b + e;
        if (i >= N) break;
        int pos = i - RANK_K + cnt[e];
        // does the id group reach a window edge?  (edge slots outside the segment never belong to it)
        const unsigned int id = id_of(mine[e]);
        const bool left_open = i - RANK_K >= 0 && id_of(WIN(e)) == id;
        const bool right_open = i + RANK_K < N && id_of(WIN(e + 2 * RANK_K)) == id;
        if (left_open || right_open) {
            // walk the whole group from memory, 8 independent loads per round trip
            int smaller = 0, first = i;
            bool go = true;
            for (int j = i - 1; go && j >= 0; j -= 8) {
                unsigned long long p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = j - u >= 0 ? pr[j - u] : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    go = go && j - u >= 0 && id_of(p[u]) == id;
                    smaller += go && p[u] < mine[e];
                    first = go ? j - u : first;
                }
            }
            go = true;
            for (int j = i + 1; go && j < N; j += 8) {
                unsigned long long p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = j + u < N ? pr[j + u] : ~0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    go = go && j + u < N && id_of(p[u]) == id;
                    smaller += go && p[u] < mine[e];
                }
            }
            pos = first + smaller;
        }
        pos_out[(size_t)seg * N + pos] = (int)(unsigned int)mine[e];
    }
#undef WIN
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
// finite min/max of one segment -> id map parameters
__global__ __launch_bounds__(SORT_THREADS) void raw_range_kernel(const float* __restrict__ keys, int L,
                                                                 SegParams* __restrict__ seg_params) {
    __shared__ float red_s[2][SORT_WAVES];
    const int tid = threadIdx.x, seg = blockIdx.x;
    const float* k = keys + (size_t)seg * L;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < L; i += SORT_THREADS) {
        const float x = k[i];
        if (x < INFINITY && x > -INFINITY) {
            lo = fminf(lo, x);
            hi = fmaxf(hi, x);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((tid & 63) == 0) { red_s[0][tid >> 6] = lo; red_s[1][tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lo = fminf(fminf(red_s[0][0], red_s[0][1]), fminf(red_s[0][2], red_s[0][3]));
        hi = fmaxf(fmaxf(red_s[1][0], red_s[1][1]), fmaxf(red_s[1][2], red_s[1][3]));
        const float width = hi - lo;
        float scale = width > 0.f ? (float)ID_BUCKETS / width : 0.f;
        if (!(scale < 3.0e38f)) scale = 0.f;
        seg_params[seg] = SegParams{lo > hi ? 0.f : lo, scale};
    }
}

// keys0 = ordered bits of the raw keys + per-chunk histogram of the low id byte
__global__ __launch_bounds__(SORT_THREADS) void raw_keygen_hist_kernel(const float* __restrict__ keys, int L,
                                                                       const SegParams* __restrict__ seg_params,
                                                                       unsigned int* __restrict__ keys0,
                                                                       unsigned int* __restrict__ hist, int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    const SegParams rg = seg_params[seg];
    h_s[tid] = 0;
    __syncthreads();
    const int base = chunk * SORT_CHUNK;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < L) {
            const unsigned int u = ordered_bits(keys[(size_t)seg * L + n]);
            keys0[(size_t)seg * L + n] = u;
            atomicAdd(&h_s[id16_of(u, rg.kmin, rg.scale) & 0xFF], 1u);
        }
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

// the passes shared by hept_sort_tables and hept_segmented_argsort: keys0 + hist(low byte) + params -> pos
struct SortBuffers {
    unsigned int* keys0;
    unsigned long long *pa, *pb;
    unsigned int* hist;
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
    b.params = reinterpret_cast<SegParams*>(ws);
    return b;
}
void run_passes(const SortBuffers& b, int segs, int N, int* pos, hipStream_t st) {
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    const dim3 grid(n_chunks, segs), block(SORT_THREADS);
    hipLaunchKernelGGL(scan_kernel, dim3(segs), dim3(RADIX), 0, st, b.hist, n_chunks);
    hipLaunchKernelGGL(scatter_kernel<false>, grid, block, 0, st, b.keys0, nullptr, b.params, b.hist, N, n_chunks, b.pa);
    hipLaunchKernelGGL(hist_hi_kernel, grid, block, 0, st, b.pa, b.params, N, b.hist, n_chunks);
    hipLaunchKernelGGL(scan_kernel, dim3(segs), dim3(RADIX), 0, st, b.hist, n_chunks);
    hipLaunchKernelGGL(scatter_kernel<true>, grid, block, 0, st, nullptr, b.pa, b.params, b.hist, N, n_chunks, b.pb);
    const dim3 grid7((N + RANK_SPAN - 1) / RANK_SPAN, segs);
    hipLaunchKernelGGL(neighbour_rank_kernel, grid7, block, 0, st, b.pb, b.params, N, pos);
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_bytes(size_t segs, size_t N) {
    const size_t n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    return align256(segs * N * 4) + 2 * align256(segs * N * 8) + align256(segs * n_chunks * RADIX * 4) +
           align256(segs * sizeof(SegParams));
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

extern "C" int hept_segmented_argsort(const float* keys, int S, int L, void* ws, int32_t* pos, void* stream) {
    if (!keys || !ws || !pos) return HEPT_ERR_ARG;
    if (S < 1 || L < 1) return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int n_chunks = (L + SORT_CHUNK - 1) / SORT_CHUNK;
    const SortBuffers b = carve_sort(ws, S, L);
    hipLaunchKernelGGL(raw_range_kernel, dim3(S), dim3(SORT_THREADS), 0, st, keys, L, b.params);
    hipLaunchKernelGGL(raw_keygen_hist_kernel, dim3(n_chunks, S), dim3(SORT_THREADS), 0, st, keys, L, b.params, b.keys0,
                       b.hist, n_chunks);
    run_passes(b, S, L, pos, st);
    return hept_launch_status();
}
