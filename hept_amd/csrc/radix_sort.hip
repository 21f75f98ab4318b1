// sort_tables: AND-shifted sort keys + stable segmented LSD radix sort.
//
// Replaces, for tables [t0, t0+Tl) (reference file:line):
//   hash_shift = max - min            example/hept_utils.py:70
//   key = hash + float(code) * shift  example/hept.py:63-65   (two rounded ops, no FMA)
//   argsort(dim=-1) x 2               example/hept.py:67-68
//
// 2*Tl*H independent segments of N keys (q segments first, then k).  Keys are sorted on
// the order-preserving u32 image of the fp32 key with 4 stable 8-bit passes, so equal keys
// keep ascending point index (= torch.sort(stable=True)); the reference's own argsort is
// unstable and leaves tie order undefined (SURVEY.md §7 hard part 1).
//
// Integer/byte work, HBM/L2-bound: per pass one histogram kernel and one scatter kernel;
// a workgroup ranks a 4096-key chunk with wave-level ballots (64-wide), no atomics on the
// data path, (key,index) travel as one 8-byte pair.
#include "common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int RADIX = 256;

__device__ __forceinline__ unsigned int ordered_bits(float key) {
    if (key == 0.f) key = 0.f;  // -0.0 and +0.0 compare equal in the reference sort
    const unsigned int u = __float_as_uint(key);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// keys0[seg][n] = ordered bits of (proj + float(code) * span); hist[seg][chunk][256] for digit 0.
__global__ __launch_bounds__(SORT_THREADS) void keygen_hist_kernel(
    const float* __restrict__ qproj, const float* __restrict__ kproj, const int64_t* __restrict__ codes,
    const float* __restrict__ minmax, int n_partials, int N, int H, int t0, int Tl,
    unsigned int* __restrict__ keys0, unsigned int* __restrict__ hist, int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    __shared__ float red_s[2][SORT_THREADS / HEPT_WAVE];
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
    h_s[tid] = 0;
    __syncthreads();
    lo = fminf(fminf(red_s[0][0], red_s[0][1]), fminf(red_s[0][2], red_s[0][3]));
    hi = fmaxf(fmaxf(red_s[1][0], red_s[1][1]), fmaxf(red_s[1][2], red_s[1][3]));
    const float span = __fsub_rn(hi, lo);

    const float* proj = (is_k ? kproj : qproj) + (size_t)th * N;
    const int64_t* code = codes + ((size_t)(t0 + t) * H + h) * N;
    unsigned int* kout = keys0 + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
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
            atomicAdd(&h_s[u & 0xFF], 1u);
        }
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

__global__ __launch_bounds__(SORT_THREADS) void hist_kernel(const unsigned long long* __restrict__ pairs, int N,
                                                            int shift, unsigned int* __restrict__ hist,
                                                            int n_chunks) {
    __shared__ unsigned int h_s[RADIX];
    const int tid = threadIdx.x, seg = blockIdx.y, chunk = blockIdx.x;
    h_s[tid] = 0;
    __syncthreads();
    const unsigned long long* src = pairs + (size_t)seg * N;
    const int base = chunk * SORT_CHUNK;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int n = base + i * SORT_THREADS + tid;
        if (n < N) atomicAdd(&h_s[(unsigned int)(src[n] >> (32 + shift)) & 0xFF], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * n_chunks + chunk) * RADIX + tid] = h_s[tid];
}

// One stable scatter pass.  FIRST: source is keys0 (index implicit); LAST: writes indices only.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(const unsigned int* __restrict__ keys0,
                                                               const unsigned long long* __restrict__ src_pairs,
                                                               unsigned long long* __restrict__ dst_pairs,
                                                               int* __restrict__ pos_out,
                                                               const unsigned int* __restrict__ hist, int N,
                                                               int n_chunks, int shift) {
    constexpr int WAVES = SORT_THREADS / HEPT_WAVE;
    __shared__ unsigned int cnt_s[WAVES][RADIX];
    __shared__ unsigned int scan_s[RADIX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.y, chunk = blockIdx.x;

    // digit `tid`: keys of this segment with a smaller digit + same digit in earlier chunks
    unsigned int before = 0, total = 0;
    {
        const unsigned int* hseg = hist + (size_t)seg * n_chunks * RADIX + tid;
        for (int c = 0; c < n_chunks; ++c) {
            const unsigned int x = hseg[(size_t)c * RADIX];
            total += x;
            if (c < chunk) before += x;
        }
    }
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) cnt_s[ww][tid] = 0;
    // exclusive scan of `total` over the 256 digits
    unsigned int incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) scan_s[w] = incl;
    __syncthreads();
    unsigned int wave_off = 0;
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww)
        if (ww < w) wave_off += scan_s[ww];
    const unsigned int digit_base = wave_off + incl - total + before;
    __syncthreads();  // scan_s is reused below

    // rank the chunk: wave w owns 1024 consecutive keys, 16 rounds of 64
    unsigned int key[SORT_ITEMS], idx[SORT_ITEMS], rank[SORT_ITEMS];
    const int wbase = chunk * SORT_CHUNK + w * (SORT_ITEMS * HEPT_WAVE);
    const size_t seg_off = (size_t)seg * N;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int n = wbase + r * HEPT_WAVE + lane;
        const bool valid = n < N;
        if (FIRST) {
            key[r] = valid ? keys0[seg_off + n] : 0xFFFFFFFFu;
            idx[r] = (unsigned int)n;
        } else {
            const unsigned long long pr = valid ? src_pairs[seg_off + n] : ~0ull;
            key[r] = (unsigned int)(pr >> 32);
            idx[r] = (unsigned int)pr;
        }
        const unsigned int dg = (key[r] >> shift) & 0xFF;
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
        rank[r] = prior + ahead;
        if (!valid) rank[r] = 0xFFFFFFFFu;
    }
    __syncthreads();
    {
        unsigned int run = digit_base;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            const unsigned int c = cnt_s[ww][tid];
            cnt_s[ww][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (rank[r] != 0xFFFFFFFFu) {
            const unsigned int dg = (key[r] >> shift) & 0xFF;
            const size_t dst = seg_off + cnt_s[w][dg] + rank[r];
            if (LAST)
                pos_out[dst] = (int)idx[r];
            else
                dst_pairs[dst] = ((unsigned long long)key[r] << 32) | idx[r];
        }
    }
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t hept_sort_workspace_bytes(int N, int H, int Tl) {
    const size_t segs = (size_t)2 * Tl * H;
    const size_t n_chunks = ((size_t)N + SORT_CHUNK - 1) / SORT_CHUNK;
    return align256(segs * N * 4) + 2 * align256(segs * N * 8) + align256(segs * n_chunks * RADIX * 4);
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
    unsigned int* keys0 = reinterpret_cast<unsigned int*>(ws);
    ws += align256((size_t)segs * N * 4);
    unsigned long long* pa = reinterpret_cast<unsigned long long*>(ws);
    ws += align256((size_t)segs * N * 8);
    unsigned long long* pb = reinterpret_cast<unsigned long long*>(ws);
    ws += align256((size_t)segs * N * 8);
    unsigned int* hist = reinterpret_cast<unsigned int*>(ws);

    const dim3 grid(n_chunks, segs), block(SORT_THREADS);
    hipLaunchKernelGGL(keygen_hist_kernel, grid, block, 0, st, qproj, kproj, codes, minmax, HEPT_PREP_GRID, N, H, t0,
                       Tl, keys0, hist, n_chunks);
    hipLaunchKernelGGL((scatter_kernel<true, false>), grid, block, 0, st, keys0, nullptr, pa, nullptr, hist, N,
                       n_chunks, 0);
    hipLaunchKernelGGL(hist_kernel, grid, block, 0, st, pa, N, 8, hist, n_chunks);
    hipLaunchKernelGGL((scatter_kernel<false, false>), grid, block, 0, st, nullptr, pa, pb, nullptr, hist, N,
                       n_chunks, 8);
    hipLaunchKernelGGL(hist_kernel, grid, block, 0, st, pb, N, 16, hist, n_chunks);
    hipLaunchKernelGGL((scatter_kernel<false, false>), grid, block, 0, st, nullptr, pb, pa, nullptr, hist, N,
                       n_chunks, 16);
    hipLaunchKernelGGL(hist_kernel, grid, block, 0, st, pa, N, 24, hist, n_chunks);
    hipLaunchKernelGGL((scatter_kernel<false, true>), grid, block, 0, st, nullptr, pa, nullptr, qpos, hist, N,
                       n_chunks, 24);
    return hept_launch_status();
}
