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
// Integer/byte work, HBM/L2-bound, four launches:
//   K1 keygen   key -> order-preserving u32 + per-chunk histogram of a MONOTONE bucket id
//               b(key) = trunc((key - kmin) * NB / (kmax - kmin)); [kmin,kmax] is the a-priori bound
//               [hash min, hash max + largest code * span] from the prep kernel's partials
//   K2 scan     histogram -> exclusive offsets [segment][chunk][bucket] (in place) + bucket starts
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
// (8192 buckets / 8192-key chunks were measured: rank -16 us, but scatter +24 us at 256 VGPRs: net loss)
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

// per-segment bucket map parameters, written once by K1 (chunk 0) and read by K3 / K4
struct SegParams {
    float kmin, scale;
};

// K1: keys0[seg][n] = ordered bits of (proj + float(code) * span); hist[seg][chunk][NB]; seg_params[seg]
__global__ __launch_bounds__(SORT_THREADS) void keygen_hist_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ minmax, int N, int H, int t0, int Tl, unsigned int* __restrict__ keys0,
    unsigned int* __restrict__ hist, SegParams* __restrict__ seg_params, int n_chunks) {
    __shared__ unsigned int h_s[NB];
    __shared__ float red_s[3][SORT_THREADS / HEPT_WAVE];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    const int th = seg % (Tl * H);  // local (table, head)
    const bool is_k = seg >= Tl * H;
    const int t = th / H, h = th % H;
#pragma unroll
    for (int i = 0; i < NB_PER_THREAD; ++i) h_s[i * SORT_THREADS + tid] = 0;

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
    // keys lie in [lo, hi + cmax*span] (codes >= 0); any key outside is clamped by bucket_of (still monotone)
    const float width = (hi + cmax * span) - lo;
    float scale = width > 0.f ? (float)NB / width : 0.f;
    if (!(scale < 3.0e38f)) scale = 0.f;  // inf/nan guard for denormal widths: one bucket, still exact
    if (chunk == 0 && tid == 0) seg_params[seg] = SegParams{lo, scale};

    const float* proj = (is_k ? kproj : qproj) + (size_t)th * N;
    const int64_t* code = codes + ((size_t)(t0 + t) * H + h) * N;
    unsigned int* kout = keys0 + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
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
            for (int e = 0; e < 4; ++e) atomicAdd(&h_s[bucket_of(u[e], lo, scale)], 1u);
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < N) {
                    const unsigned int u = make_key(proj[n + e], code[n + e]);
                    kout[n + e] = u;
                    atomicAdd(&h_s[bucket_of(u, lo, scale)], 1u);
                }
        }
    }
    __syncthreads();
    unsigned int* dst = hist + ((size_t)seg * n_chunks + chunk) * NB;
#pragma unroll
    for (int i = 0; i < NB_PER_THREAD; ++i) dst[i * SORT_THREADS + tid] = h_s[i * SORT_THREADS + tid];
}

// K2: one workgroup per segment.  hist[seg][c][b] <- start[b] + sum_{c' < c} hist[seg][c'][b];
// bucket_start[seg][b] = start[b] = number of keys in smaller buckets; bucket_start[seg][NB] = N.
__global__ __launch_bounds__(SCAN_THREADS) void bucket_scan_kernel(unsigned int* __restrict__ hist, int n_chunks,
                                                                   unsigned int* __restrict__ bucket_start) {
    constexpr int PER = NB / SCAN_THREADS;  // consecutive buckets per thread
    constexpr int V = PER / 4;
    constexpr int WAVES = SCAN_THREADS / HEPT_WAVE;
    constexpr int BATCH = 8;                // chunk rows in flight per thread
    static_assert(PER % 4 == 0, "bucket count must be a multiple of 4 * SCAN_THREADS");
    __shared__ unsigned int wsum_s[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, seg = blockIdx.x;
    unsigned int* hseg = hist + (size_t)seg * n_chunks * NB + tid * PER;
    u32x4 total[V];
#pragma unroll
    for (int j = 0; j < V; ++j) total[j] = u32x4{0u, 0u, 0u, 0u};
    for (int c0 = 0; c0 < n_chunks; c0 += BATCH) {
        u32x4 x[BATCH][V];
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j)
                x[i][j] = (c0 + i < n_chunks) ? *reinterpret_cast<const u32x4*>(hseg + (size_t)(c0 + i) * NB + 4 * j)
                                              : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j) total[j] += x[i][j];
    }
    unsigned int mine = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) mine += total[j][0] + total[j][1] + total[j][2] + total[j][3];
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
    u32x4 acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[j][i] = run;
            run += total[j][i];
        }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        unsigned int* bs = bucket_start + (size_t)seg * (NB + 1) + tid * PER + 4 * j;
        bs[0] = acc[j][0]; bs[1] = acc[j][1]; bs[2] = acc[j][2]; bs[3] = acc[j][3];  // (NB+1)-pitch rows: 4-B aligned only
    }
    if (tid == SCAN_THREADS - 1) bucket_start[(size_t)seg * (NB + 1) + NB] = run;
    for (int c0 = 0; c0 < n_chunks; c0 += BATCH) {
        u32x4 x[BATCH][V];
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j)
                x[i][j] = (c0 + i < n_chunks) ? *reinterpret_cast<const u32x4*>(hseg + (size_t)(c0 + i) * NB + 4 * j)
                                              : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (c0 + i < n_chunks) *reinterpret_cast<u32x4*>(hseg + (size_t)(c0 + i) * NB + 4 * j) = acc[j];
                acc[j] += x[i][j];
            }
    }
}

// K3: stable scatter by bucket id -> (key,index) pairs in bucket order
__global__ __launch_bounds__(SORT_THREADS) void bucket_scatter_kernel(
    const unsigned int* __restrict__ keys0, const SegParams* __restrict__ seg_params,
    const unsigned int* __restrict__ offs, int N, int n_chunks, unsigned long long* __restrict__ pairs) {
    constexpr int WAVES = SORT_THREADS / HEPT_WAVE;
    __shared__ __attribute__((aligned(16))) unsigned short cnt_s[WAVES][NB];  // per-wave bucket counters (<= 1024 keys per wave)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const SegParams rg = seg_params[seg];
    // this wave's counters only: no workgroup barrier needed before ranking
    {
        u32x4* z = reinterpret_cast<u32x4*>(&cnt_s[w][0]);
#pragma unroll
        for (int i = 0; i < NB * 2 / 16 / HEPT_WAVE; ++i) z[i * HEPT_WAVE + lane] = u32x4{0u, 0u, 0u, 0u};
    }

    // rank the chunk: wave w owns 1024 consecutive keys, 16 rounds of 64 (stable: index order)
    unsigned int key[SORT_ITEMS];
    unsigned short rank[SORT_ITEMS], dig[SORT_ITEMS];
    const int wbase = chunk * SORT_CHUNK + w * (SORT_ITEMS * HEPT_WAVE);
    const size_t seg_off = (size_t)seg * N;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        key[r] = n < N ? keys0[seg_off + n] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        const bool valid = n < N;
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
    const unsigned int* off_c = offs + ((size_t)seg * n_chunks + chunk) * NB;
    unsigned int base_r[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) base_r[r] = off_c[dig[r]];
    __syncthreads();
    // keys of the same bucket held by earlier waves of this workgroup come first (only the touched
    // counters are read: the chunk has as many keys as there are buckets)
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        unsigned int before_w = 0;
#pragma unroll
        for (int ww = 0; ww < WAVES - 1; ++ww)
            if (ww < w) before_w += cnt_s[ww][dig[r]];
        const int n = wbase + r * HEPT_WAVE + lane;
        if (n < N) {
            const size_t dst = seg_off + base_r[r] + before_w + rank[r];
            pairs[dst] = ((unsigned long long)key[r] << 32) | (unsigned int)n;
        }
    }
}

// K4: final position = bucket start + number of pairs of the same bucket that are smaller.  A pair is
// (key << 32 | index); inside a bucket the scatter kept ascending index order, so u64 order of the pairs
// is exactly (key, then original index): the stable tie rule.  The workgroup's 256 positions plus the
// rest of their first/last buckets are staged in LDS (same-address reads broadcast within a wave).
constexpr int RANK_PER_THREAD = 4;
constexpr int RANK_SPAN = SORT_THREADS * RANK_PER_THREAD;  // positions per workgroup
constexpr int RANK_CAP = 2048;
__global__ __launch_bounds__(SORT_THREADS) void bucket_rank_kernel(const unsigned long long* __restrict__ pairs,
                                                                   const SegParams* __restrict__ seg_params,
                                                                   const unsigned int* __restrict__ bucket_start,
                                                                   int N, int* __restrict__ pos_out) {
    __shared__ unsigned long long p_s[RANK_CAP];
    __shared__ int range_s[2];
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int i0 = blockIdx.x * RANK_SPAN;
    const int last = min(i0 + RANK_SPAN, N) - 1;
    const SegParams rg = seg_params[seg];
    const unsigned long long* pr = pairs + (size_t)seg * N;
    const unsigned int* bs = bucket_start + (size_t)seg * (NB + 1);
    unsigned long long mine[RANK_PER_THREAD];
    int s[RANK_PER_THREAD], e[RANK_PER_THREAD];
#pragma unroll
    for (int u = 0; u < RANK_PER_THREAD; ++u) {
        const int i = i0 + u * SORT_THREADS + tid;
        mine[u] = i < N ? pr[i] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < RANK_PER_THREAD; ++u) {
        const int i = i0 + u * SORT_THREADS + tid;
        s[u] = 0;
        e[u] = 0;
        if (i < N) {
            const int b = bucket_of((unsigned int)(mine[u] >> 32), rg.kmin, rg.scale);
            s[u] = (int)bs[b];
            e[u] = (int)bs[b + 1];
        }
        if (i == i0) range_s[0] = s[u];
        if (i == last) range_s[1] = e[u];
    }
    __syncthreads();
    const int lo = range_s[0], hi = range_s[1];
    int smaller[RANK_PER_THREAD] = {0, 0, 0, 0};
    if (hi - lo <= RANK_CAP) {
        for (int j = lo + tid; j < hi; j += SORT_THREADS) p_s[j - lo] = pr[j];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RANK_PER_THREAD; ++u) {
#pragma unroll 4
            for (int j = s[u]; j < e[u]; ++j) smaller[u] += p_s[j - lo] < mine[u];
        }
    } else {
        // oversized bucket(s): same count straight from memory (slow path, adversarial inputs only)
#pragma unroll
        for (int u = 0; u < RANK_PER_THREAD; ++u)
            for (int j = s[u]; j < e[u]; ++j) smaller[u] += pr[j] < mine[u];
    }
#pragma unroll
    for (int u = 0; u < RANK_PER_THREAD; ++u) {
        const int i = i0 + u * SORT_THREADS + tid;
        if (i < N) pos_out[(size_t)seg * N + s[u] + smaller[u]] = (int)(unsigned int)mine[u];
    }
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t hept_sort_workspace_bytes(int N, int H, int Tl) {
    const size_t segs = (size_t)2 * Tl * H;
    const size_t n_chunks = ((size_t)N + SORT_CHUNK - 1) / SORT_CHUNK;
    return align256(segs * N * 4) + align256(segs * N * 8) + align256(segs * n_chunks * NB * 4) +
           align256(segs * sizeof(SegParams)) + align256(segs * (NB + 1) * 4);
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
    SegParams* params = reinterpret_cast<SegParams*>(take((size_t)segs * sizeof(SegParams)));
    unsigned int* bstart = reinterpret_cast<unsigned int*>(take((size_t)segs * (NB + 1) * 4));

    const dim3 grid(n_chunks, segs), block(SORT_THREADS);
    hipLaunchKernelGGL(keygen_hist_kernel, grid, block, 0, st, qproj, kproj, codes, minmax, N, H, t0, Tl, keys0, hist,
                       params, n_chunks);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(segs), dim3(SCAN_THREADS), 0, st, hist, n_chunks, bstart);
    hipLaunchKernelGGL(bucket_scatter_kernel, grid, block, 0, st, keys0, params, hist, N, n_chunks, pairs);
    const dim3 grid4((N + RANK_SPAN - 1) / RANK_SPAN, segs);
    hipLaunchKernelGGL(bucket_rank_kernel, grid4, block, 0, st, pairs, params, bstart, N, qpos);
    return hept_launch_status();
}
