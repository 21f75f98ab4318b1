// sort_tables: AND-shifted sort keys + stable segmented sort (bucket pass + in-bucket ranking).
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
// Integer/byte work, HBM/L2-bound, five launches:
//   K1 keygen   key -> order-preserving u32, per-chunk min/max
//   K2 hist     per-chunk histogram of a MONOTONE bucket id b(key) = trunc((key-kmin)*NB/(kmax-kmin))
//   K2b scan    histogram -> exclusive offsets [segment][chunk][bucket] (in place) + bucket starts
//   K3 scatter  stable counting-sort pass on b (wave-level ballots, 64-wide; no data-path atomics),
//               (key,index) travel as one 8-byte pair
//   K4 rank     every element counts the smaller keys of its own bucket -> final position
// b() is a monotone non-decreasing function of the key itself (fp32 sub, mul and truncation are all
// monotone), so bucket order never contradicts key order and ties never straddle buckets: the
// two-level result is the exact stable sort for ANY input.  Cost is O(N + sum bucket^2): with the
// quantile AND codes of HEPT the keys are near-uniform over [kmin,kmax] (~N/4096 per bucket);
// adversarial inputs (all keys within 1/4096 of the range) degrade to O(N^2) compares per
// segment — slow, never wrong.
#include "common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int NB = 4096;                               // buckets per segment (12-bit digit)
constexpr int NB_BITS = 12;
constexpr int NB_PER_THREAD = NB / SORT_THREADS;       // 16
constexpr int SCAN_THREADS = 1024;

__device__ __forceinline__ unsigned int ordered_bits(float key) {
    if (key == 0.f) key = 0.f;  // -0.0 and +0.0 compare equal in the reference sort
    const unsigned int u = __float_as_uint(key);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned int u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// monotone bucket id; scale = NB / (kmax - kmin), 0 when all keys are equal
__device__ __forceinline__ int bucket_of(unsigned int u, float kmin, float scale) {
    const float x = (from_ordered(u) - kmin) * scale;
    const int b = (int)x;
    return b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
}

struct SegRange {
    float kmin, scale;
};
// reduce the per-chunk [min,max] of a segment (written by K1); every thread gets the same answer
__device__ __forceinline__ SegRange segment_range(const unsigned int* __restrict__ chunk_mm, int seg, int n_chunks) {
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (int c = 0; c < n_chunks; ++c) {
        lo = min(lo, chunk_mm[((size_t)seg * n_chunks + c) * 2]);
        hi = max(hi, chunk_mm[((size_t)seg * n_chunks + c) * 2 + 1]);
    }
    SegRange r;
    r.kmin = from_ordered(lo);
    const float width = from_ordered(hi) - r.kmin;
    r.scale = width > 0.f ? (float)NB / width : 0.f;
    if (!(r.scale < 3.0e38f)) r.scale = 0.f;  // inf/nan guard for denormal widths: one bucket, still exact
    return r;
}

// K1: keys0[seg][n] = ordered bits of (proj + float(code) * span); chunk_mm[seg][chunk] = [min,max]
__global__ __launch_bounds__(SORT_THREADS) void keygen_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ minmax, int n_partials, int N, int H, int t0, int Tl,
    unsigned int* __restrict__ keys0, unsigned int* __restrict__ chunk_mm, int n_chunks) {
    __shared__ float red_s[2][SORT_THREADS / HEPT_WAVE];
    __shared__ unsigned int redu_s[2][SORT_THREADS / HEPT_WAVE];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    const int th = seg % (Tl * H);  // local (table, head)
    const bool is_k = seg >= Tl * H;
    const int t = th / H, h = th % H;

    // hash range of this (table, head): reduce the prep kernel's per-workgroup partials
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < n_partials; i += SORT_THREADS) {
        const float* m = minmax + (((size_t)i * Tl + t) * H + h) * 2;
        lo = fminf(lo, m[0]);
        hi = fmaxf(hi, m[1]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((tid & 63) == 0) { red_s[0][tid >> 6] = lo; red_s[1][tid >> 6] = hi; }
    __syncthreads();
    lo = fminf(fminf(red_s[0][0], red_s[0][1]), fminf(red_s[0][2], red_s[0][3]));
    hi = fmaxf(fmaxf(red_s[1][0], red_s[1][1]), fmaxf(red_s[1][2], red_s[1][3]));
    const float span = hi - lo;

    const float* proj = (is_k ? kproj : qproj) + (size_t)th * N;
    const int64_t* code = codes + ((size_t)(t0 + t) * H + h) * N;
    unsigned int* kout = keys0 + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
    unsigned int umin = 0xFFFFFFFFu, umax = 0u;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < N) {
            // two separately rounded ops, as the two eager ops of the reference; HIP's __fmul_rn /
            // __fadd_rn are plain * and + and would be contracted to one fma without
            // -ffp-contract=off (Makefile) -- the asm barrier makes it explicit here as well
            float off = (float)code[n] * span;
            asm volatile("" : "+v"(off));
            const unsigned int u = ordered_bits(proj[n] + off);
            kout[n] = u;
            umin = min(umin, u);
            umax = max(umax, u);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        umin = min(umin, (unsigned int)__shfl_xor((int)umin, off));
        umax = max(umax, (unsigned int)__shfl_xor((int)umax, off));
    }
    if ((tid & 63) == 0) { redu_s[0][tid >> 6] = umin; redu_s[1][tid >> 6] = umax; }
    __syncthreads();
    if (tid == 0) {
        chunk_mm[((size_t)seg * n_chunks + chunk) * 2] = min(min(redu_s[0][0], redu_s[0][1]), min(redu_s[0][2], redu_s[0][3]));
        chunk_mm[((size_t)seg * n_chunks + chunk) * 2 + 1] =
            max(max(redu_s[1][0], redu_s[1][1]), max(redu_s[1][2], redu_s[1][3]));
    }
}

// K2: hist[seg][chunk][NB] of the bucket id
__global__ __launch_bounds__(SORT_THREADS) void bucket_hist_kernel(const unsigned int* __restrict__ keys0,
                                                                   const unsigned int* __restrict__ chunk_mm, int N,
                                                                   unsigned int* __restrict__ hist, int n_chunks) {
    __shared__ unsigned int h_s[NB];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
#pragma unroll
    for (int i = 0; i < NB_PER_THREAD; ++i) h_s[i * SORT_THREADS + tid] = 0;
    const SegRange rg = segment_range(chunk_mm, seg, n_chunks);
    __syncthreads();
    const unsigned int* src = keys0 + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < N) atomicAdd(&h_s[bucket_of(src[n], rg.kmin, rg.scale)], 1u);
    }
    __syncthreads();
    unsigned int* dst = hist + ((size_t)seg * n_chunks + chunk) * NB;
#pragma unroll
    for (int i = 0; i < NB_PER_THREAD; ++i) dst[i * SORT_THREADS + tid] = h_s[i * SORT_THREADS + tid];
}

// K2b: one workgroup per segment.  hist[seg][c][b] <- start[b] + sum_{c' < c} hist[seg][c'][b];
// bucket_start[seg][b] = start[b] = number of keys in smaller buckets; bucket_start[seg][NB] = N.
__global__ __launch_bounds__(SCAN_THREADS) void bucket_scan_kernel(unsigned int* __restrict__ hist, int n_chunks,
                                                                   unsigned int* __restrict__ bucket_start) {
    constexpr int PER = NB / SCAN_THREADS;  // 4 consecutive buckets per thread
    constexpr int WAVES = SCAN_THREADS / HEPT_WAVE;
    __shared__ unsigned int wsum_s[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, seg = blockIdx.x;
    unsigned int* hseg = hist + (size_t)seg * n_chunks * NB + tid * PER;
    unsigned int total[PER] = {0, 0, 0, 0};
    for (int c = 0; c < n_chunks; ++c) {
        const u32x4 x = *reinterpret_cast<const u32x4*>(hseg + (size_t)c * NB);
#pragma unroll
        for (int i = 0; i < PER; ++i) total[i] += x[i];
    }
    const unsigned int mine = total[0] + total[1] + total[2] + total[3];
    unsigned int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum_s[w] = incl;
    __syncthreads();
    unsigned int run = incl - mine;
    for (int ww = 0; ww < w; ++ww) run += wsum_s[ww];
    u32x4 acc;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        acc[i] = run;
        run += total[i];
    }
    *reinterpret_cast<u32x4*>(bucket_start + (size_t)seg * (NB + 1) + tid * PER) = acc;
    if (tid == SCAN_THREADS - 1) bucket_start[(size_t)seg * (NB + 1) + NB] = run;
    for (int c = 0; c < n_chunks; ++c) {
        u32x4* p = reinterpret_cast<u32x4*>(hseg + (size_t)c * NB);
        const u32x4 x = *p;
        *p = acc;
        acc += x;
    }
}

// K3: stable scatter by bucket id -> (key,index) pairs in bucket order
__global__ __launch_bounds__(SORT_THREADS) void bucket_scatter_kernel(
    const unsigned int* __restrict__ keys0, const unsigned int* __restrict__ chunk_mm,
    const unsigned int* __restrict__ offs, int N, int n_chunks, unsigned long long* __restrict__ pairs) {
    constexpr int WAVES = SORT_THREADS / HEPT_WAVE;
    __shared__ unsigned short cnt_s[WAVES][NB];  // per-wave bucket counters (<= 1024 keys per wave)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const SegRange rg = segment_range(chunk_mm, seg, n_chunks);
    {
        unsigned int* z = reinterpret_cast<unsigned int*>(&cnt_s[0][0]);
        for (int i = tid; i < WAVES * NB / 2; i += SORT_THREADS) z[i] = 0;
    }
    __syncthreads();

    // rank the chunk: wave w owns 1024 consecutive keys, 16 rounds of 64 (stable: index order)
    unsigned int key[SORT_ITEMS];
    unsigned short rank[SORT_ITEMS], dig[SORT_ITEMS];
    const int wbase = chunk * SORT_CHUNK + w * (SORT_ITEMS * HEPT_WAVE);
    const size_t seg_off = (size_t)seg * N;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        const bool valid = n < N;
        key[r] = valid ? keys0[seg_off + n] : 0xFFFFFFFFu;
        const unsigned int dg = valid ? (unsigned int)bucket_of(key[r], rg.kmin, rg.scale) : (unsigned int)(NB - 1);
        dig[r] = (unsigned short)dg;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < NB_BITS; ++b) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const unsigned int prior = cnt_s[w][dg];
        const unsigned int ahead = __popcll(peers & lt_mask);
        if (valid && ahead == 0) cnt_s[w][dg] = (unsigned short)(prior + __popcll(peers));
        rank[r] = (unsigned short)(prior + ahead);
    }
    __syncthreads();
    // per-wave counters -> exclusive prefix over the waves of this workgroup
#pragma unroll
    for (int i = 0; i < NB_PER_THREAD; ++i) {
        const int d = i * SORT_THREADS + tid;
        unsigned int acc = 0;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            const unsigned int c = cnt_s[ww][d];
            cnt_s[ww][d] = (unsigned short)acc;
            acc += c;
        }
    }
    __syncthreads();
    const unsigned int* off_c = offs + ((size_t)seg * n_chunks + chunk) * NB;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        if (n < N) {
            const size_t dst = seg_off + off_c[dig[r]] + cnt_s[w][dig[r]] + rank[r];
            pairs[dst] = ((unsigned long long)key[r] << 32) | (unsigned int)n;
        }
    }
}

// K4: final position = bucket start + number of keys of the same bucket that sort before this one
__global__ __launch_bounds__(SORT_THREADS) void bucket_rank_kernel(const unsigned long long* __restrict__ pairs,
                                                                   const unsigned int* __restrict__ chunk_mm,
                                                                   const unsigned int* __restrict__ bucket_start,
                                                                   int N, int n_chunks, int* __restrict__ pos_out) {
    const int seg = blockIdx.y;
    const int i = blockIdx.x * SORT_THREADS + threadIdx.x;
    const SegRange rg = segment_range(chunk_mm, seg, n_chunks);
    if (i >= N) return;
    const uint2* pr = reinterpret_cast<const uint2*>(pairs + (size_t)seg * N);  // .x = index, .y = key
    const uint2 mine = pr[i];
    const unsigned int me = mine.y;
    const int b = bucket_of(me, rg.kmin, rg.scale);
    const unsigned int* bs = bucket_start + (size_t)seg * (NB + 1);
    const int s = (int)bs[b], e = (int)bs[b + 1];
    // keys of one bucket are contiguous; lanes of a wave mostly share a bucket, so the loads broadcast.
    // j < i in bucket order <=> smaller original index (the scatter pass is stable).
    int smaller = 0;
#pragma unroll 4
    for (int j = s; j < e; ++j) {
        const unsigned int kj = pr[j].y;
        smaller += (kj < me) || (kj == me && j < i);
    }
    pos_out[(size_t)seg * N + s + smaller] = (int)mine.x;
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t hept_sort_workspace_bytes(int N, int H, int Tl) {
    const size_t segs = (size_t)2 * Tl * H;
    const size_t n_chunks = ((size_t)N + SORT_CHUNK - 1) / SORT_CHUNK;
    return align256(segs * N * 4) + align256(segs * N * 8) + align256(segs * n_chunks * NB * 4) +
           align256(segs * n_chunks * 2 * 4) + align256(segs * (NB + 1) * 4);
}

extern "C" int hept_sort_tables(const float* qproj, const float* kproj, const int64_t* codes, const float* minmax,
                                int N, int H, int T, int t0, int Tl, void* sort_ws, int32_t* qpos, int32_t* kpos,
                                void* stream) {
    if (!qproj || !kproj || !codes || !minmax || !sort_ws || !qpos || !kpos) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || Tl > HEPT_MAX_TABLES || t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (kpos != qpos + (size_t)Tl * H * N) return HEPT_ERR_ARG;  // one (2,Tl,H,N) array: q then k
    hipStream_t st = (hipStream_t)stream;
    const int segs = 2 * Tl * H;
    const int n_chunks = (N + SORT_CHUNK - 1) / SORT_CHUNK;
    char* ws = reinterpret_cast<char*>(sort_ws);
    auto take = [&](size_t bytes) {
        char* r = ws;
        ws += align256(bytes);
        return r;
    };
    unsigned int* keys0 = reinterpret_cast<unsigned int*>(take((size_t)segs * N * 4));
    unsigned long long* pairs = reinterpret_cast<unsigned long long*>(take((size_t)segs * N * 8));
    unsigned int* hist = reinterpret_cast<unsigned int*>(take((size_t)segs * n_chunks * NB * 4));
    unsigned int* chunk_mm = reinterpret_cast<unsigned int*>(take((size_t)segs * n_chunks * 2 * 4));
    unsigned int* bstart = reinterpret_cast<unsigned int*>(take((size_t)segs * (NB + 1) * 4));

    const dim3 grid(n_chunks, segs), block(SORT_THREADS);
    hipLaunchKernelGGL(keygen_kernel, grid, block, 0, st, qproj, kproj, codes, minmax, HEPT_PREP_GRID, N, H, t0, Tl,
                       keys0, chunk_mm, n_chunks);
    hipLaunchKernelGGL(bucket_hist_kernel, grid, block, 0, st, keys0, chunk_mm, N, hist, n_chunks);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(segs), dim3(SCAN_THREADS), 0, st, hist, n_chunks, bstart);
    hipLaunchKernelGGL(bucket_scatter_kernel, grid, block, 0, st, keys0, chunk_mm, hist, N, n_chunks, pairs);
    const dim3 grid4((N + SORT_THREADS - 1) / SORT_THREADS, segs);
    hipLaunchKernelGGL(bucket_rank_kernel, grid4, block, 0, st, pairs, chunk_mm, bstart, N, n_chunks, qpos);
    return hept_launch_status();
}
