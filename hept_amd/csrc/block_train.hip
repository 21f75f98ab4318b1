// block_train: the small dense pieces around the operator in the TRAINING step of the Attn block (SURVEY.md §8 f-4 x
// f-2; reference example/transformer.py:154-165 under autograd, example/trainer.py:11-22).
//
// The reference composes LayerNorm / Linear / ReLU modules and lets autograd do the rest.  On (60 000, 24) activations
// that is ~1 ms per block and step of library kernels that were not made for this shape: the five weight gradients
// dW = dY^T . X are (192 or 24) x 24 outputs reduced over 60 000 points -- rocBLAS runs them at 138-156 us EACH -- and
// the LayerNorm forward / backward kernels take 59 + 51 + 31 us per norm for 5.8 MB.  Here:
//   hept_rows_wgrad   dW[o][j] = sum_n dY[n][o] X[n][j], db[o] = sum_n dY[n][o]          (any Linear(24 -> O))
//   hept_ln_bwd       LayerNorm(24) backward + the normalised rows (norm1 in front of the fused row builder)
//   hept_ln_ffn_fwd   out = ff.2(relu(ff.0(norm2(x1))))                                  (:162 without the residual)
//   hept_ln_ffn_bwd   its backward: d x1, the rows the two weight gradients need, LayerNorm parameter partials
// Every reduction over the points is two-stage with a fixed association (per-workgroup partials, then hept_fixed_sum):
// two backward passes over the same inputs are bit-identical, like the rest of the training path.
#include "common.h"

namespace {

constexpr int BT_D = 24;            // row width of the block's activations (h_dim of the shipped models)
constexpr int BT_POINTS = 128;      // points per workgroup of the reductions

__device__ __forceinline__ float wave_sum_f(float x) {   // the same value in every lane; fixed order (DPP scan)
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, true));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// ---- weight gradient of a Linear(24 -> O): partial[wg][c][o] = sum over the workgroup's points of dY[n][o] * X[n][c]
// (c < 24) and of dY[n][o] (c = 24: the bias gradient).  Thread = column o of one of GROUPS thread groups; the groups
// take every GROUPS-th point (4 points in flight per thread: the loop is bound by load latency, not by its 24 fma per
// point), the workgroup's X rows are LDS broadcasts, the groups fold through LDS.
template <int COLS, int GROUPS>
__global__ __launch_bounds__(COLS * GROUPS) void wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                               int N, int O, float* __restrict__ partial) {
    __shared__ float red_s[GROUPS - 1][BT_D + 1][COLS];
    static_assert(sizeof(float) * BT_POINTS * BT_D <= sizeof(float) * (GROUPS - 1) * (BT_D + 1) * COLS, "x tile fits");
    float* x_s = &red_s[0][0][0];   // the workgroup's X rows during the loop, then the fold of the thread groups
    const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    const int o = blockIdx.y * COLS + col;
    const bool live = o < O;
    const int n_begin = blockIdx.x * BT_POINTS, n_end = min(N, n_begin + BT_POINTS);
    float s[BT_D];
#pragma unroll
    for (int c = 0; c < BT_D; ++c) s[c] = 0.f;
    float sb = 0.f;
    for (int i = threadIdx.x; i < (n_end - n_begin) * BT_D; i += COLS * GROUPS) x_s[i] = X[(size_t)n_begin * BT_D + i];
    __syncthreads();
    constexpr int UN = 4;
    for (int n0 = n_begin + grp; n0 < n_end; n0 += GROUPS * UN) {
        float dy[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + u * GROUPS;
            dy[u] = (live && n < n_end) ? dY[(size_t)n * O + o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + u * GROUPS;
            if (n < n_end) {                                    // uniform over the thread group
                const float* xr = x_s + (n - n_begin) * BT_D;   // the same word in every lane: LDS broadcast
#pragma unroll
                for (int c = 0; c < BT_D; ++c) s[c] = fmaf(xr[c], dy[u], s[c]);
                sb += dy[u];
            }
        }
    }
    __syncthreads();  // every group is done with the x tile
    if (grp > 0) {
#pragma unroll
        for (int c = 0; c < BT_D; ++c) red_s[grp - 1][c][col] = s[c];
        red_s[grp - 1][BT_D][col] = sb;
    }
    __syncthreads();
    if (grp == 0 && live) {
#pragma unroll
        for (int g2 = 0; g2 < GROUPS - 1; ++g2) {   // groups added in index order
#pragma unroll
            for (int c = 0; c < BT_D; ++c) s[c] += red_s[g2][c][col];
            sb += red_s[g2][BT_D][col];
        }
        float* mine = partial + (size_t)blockIdx.x * (BT_D + 1) * O;   // [c][o], row BT_D = bias sums
#pragma unroll
        for (int c = 0; c < BT_D; ++c) mine[(size_t)c * O + o] = s[c];
        mine[(size_t)BT_D * O + o] = sb;
    }
}

// second stage: dW[o][c] = sum over workgroups of partial[wg][c][o] (c < 24), db[o] from row 24; fixed association
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float* __restrict__ partial, int n_wgs, int O,
                                                        float* __restrict__ d_weight, float* __restrict__ d_bias) {
    __shared__ float red_s[HEPT_FSUM_SLICES * HEPT_FSUM_OUT];
    const int total = (BT_D + 1) * O;
    const int i = blockIdx.x * HEPT_FSUM_OUT + threadIdx.x % HEPT_FSUM_OUT;
    const bool valid = i < total;
    const float tot = hept_fixed_sum(partial, n_wgs, (size_t)total, i, valid, red_s);
    if (threadIdx.x < HEPT_FSUM_OUT && valid) {
        const int c = i / O, o = i - c * O;
        if (c < BT_D) d_weight[(size_t)o * BT_D + c] = tot;
        else if (d_bias) d_bias[o] = tot;
    }
}

// generic column sums of per-workgroup partial rows: out[i] = sum_wg partial[wg][i], i < width (LayerNorm parameters)
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ partial, int n_wgs, int width,
                                                          float* __restrict__ out_a, float* __restrict__ out_b, int split) {
    __shared__ float red_s[HEPT_FSUM_SLICES * HEPT_FSUM_OUT];
    const int i = blockIdx.x * HEPT_FSUM_OUT + threadIdx.x % HEPT_FSUM_OUT;
    const bool valid = i < width;
    const float tot = hept_fixed_sum(partial, n_wgs, (size_t)width, i, valid, red_s);
    if (threadIdx.x < HEPT_FSUM_OUT && valid) {
        if (i < split) out_a[i] = tot;
        else out_b[i - split] = tot;
    }
}

// ---- per-point pieces (one lane = one point; a row is 24 floats = six 16-B loads, consecutive lanes consecutive rows)
struct Row {
    float v[BT_D];
    __device__ __forceinline__ void load(const float* __restrict__ p) {
        const f32x4* s = reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int j = 0; j < BT_D / 4; ++j) {
            const f32x4 q = s[j];
            v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ p) const {
        f32x4* d = reinterpret_cast<f32x4*>(p);
#pragma unroll
        for (int j = 0; j < BT_D / 4; ++j) d[j] = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    }
};

// LayerNorm(24) of a row as torch computes it (biased variance, eps inside the root): xhat, and mean / rstd
__device__ __forceinline__ void ln_stats(const Row& x, float eps, Row& xhat, float& rstd) {
    float mean = 0.f;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) mean += x.v[j];
    mean *= 1.0f / BT_D;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) { const float d = x.v[j] - mean; var = fmaf(d, d, var); }
    rstd = 1.0f / sqrtf(var * (1.0f / BT_D) + eps);
#pragma unroll
    for (int j = 0; j < BT_D; ++j) xhat.v[j] = (x.v[j] - mean) * rstd;
}
// d x of LayerNorm from d(xhat * w + b) = dz:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dz * w
__device__ __forceinline__ void ln_backward(const Row& dz, const Row& xhat, float rstd, const float* __restrict__ w_s,
                                            Row& dx) {
    float g[BT_D], m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) {
        g[j] = dz.v[j] * w_s[j];
        m1 += g[j];
        m2 = fmaf(g[j], xhat.v[j], m2);
    }
    m1 *= 1.0f / BT_D;
    m2 *= 1.0f / BT_D;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) dx.v[j] = rstd * (g[j] - m1 - xhat.v[j] * m2);
}

// per-workgroup partial sums of the LayerNorm parameter gradients: partial[wg][0..23] = sum dz * xhat (weight),
// [24..47] = sum dz (bias); waves by DPP, then the waves in index order
template <int NT>
__device__ __forceinline__ void ln_param_partials(const Row& dz, const Row& xhat, bool valid, float* __restrict__ wave_s,
                                                  float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) {
        const float a = wave_sum_f(valid ? dz.v[j] * xhat.v[j] : 0.f);
        const float b = wave_sum_f(valid ? dz.v[j] : 0.f);
        if (lane == 0) { wave_s[w * 2 * BT_D + j] = a; wave_s[w * 2 * BT_D + BT_D + j] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * BT_D) {
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < NT / 64; ++ww) tot += wave_s[ww * 2 * BT_D + threadIdx.x];
        partial[(size_t)blockIdx.x * 2 * BT_D + threadIdx.x] = tot;
    }
}

constexpr int LN_THREADS = 256;

// norm1 backward in front of the fused row builder: dx, the normalised rows xn (what the three weight gradients
// multiply), parameter partials
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dxn,
                                                            const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                            float eps, int N, float* __restrict__ dx,
                                                            float* __restrict__ xn_out, float* __restrict__ partial,
                                                            float* __restrict__ dump) {
    __shared__ float w_s[2 * BT_D];
    __shared__ float wave_s[(LN_THREADS / 64) * 2 * BT_D];
    if (threadIdx.x < BT_D) { w_s[threadIdx.x] = ln_w[threadIdx.x]; w_s[BT_D + threadIdx.x] = ln_b[threadIdx.x]; }
    __syncthreads();
    const int n = blockIdx.x * LN_THREADS + threadIdx.x;
    const bool valid = n < N;
    const size_t row = (size_t)(valid ? n : N - 1) * BT_D;
    Row xr, dz, xhat, out;
    xr.load(x + row);
    dz.load(dxn + row);
    float rstd;
    ln_stats(xr, eps, xhat, rstd);
    ln_backward(dz, xhat, rstd, w_s, out);
    // (lanes past the last point store into a dump row: a branch around the stores makes the compiler copy the whole
    //  register rows per store -- 256 VGPRs and scratch in the ffn kernel below)
    out.store(valid ? dx + row : dump);
#pragma unroll
    for (int j = 0; j < BT_D; ++j) out.v[j] = fmaf(xhat.v[j], w_s[j], w_s[BT_D + j]);
    out.store(valid ? xn_out + row : dump);
    ln_param_partials<LN_THREADS>(dz, xhat, valid, wave_s, partial);
}

struct FfnW {
    const float *ln_w, *ln_b, *w1, *b1, *w2, *b2;
    float eps;
};
constexpr int FFN_W = 2 * BT_D * BT_D + 4 * BT_D;   // [w1 | w2 | b1 | b2 | ln_w | ln_b]
__device__ __forceinline__ void stage_ffn(const FfnW& p, float* __restrict__ s) {
    for (int i = threadIdx.x; i < BT_D * BT_D; i += blockDim.x) {
        s[i] = p.w1[i];
        s[BT_D * BT_D + i] = p.w2[i];
    }
    if (threadIdx.x < BT_D) {
        s[2 * BT_D * BT_D + threadIdx.x] = p.b1[threadIdx.x];
        s[2 * BT_D * BT_D + BT_D + threadIdx.x] = p.b2[threadIdx.x];
        s[2 * BT_D * BT_D + 2 * BT_D + threadIdx.x] = p.ln_w[threadIdx.x];
        s[2 * BT_D * BT_D + 3 * BT_D + threadIdx.x] = p.ln_b[threadIdx.x];
    }
    __syncthreads();
}
// z = norm2(x1); a = ff.0(z); (weights are LDS broadcasts: every lane reads the same word)
__device__ __forceinline__ void ffn_hidden(const Row& xhat, const float* __restrict__ s, Row& z, Row& a) {
    const float* lw = s + 2 * BT_D * BT_D + 2 * BT_D;
#pragma unroll
    for (int j = 0; j < BT_D; ++j) z.v[j] = fmaf(xhat.v[j], lw[j], lw[BT_D + j]);
#pragma unroll
    for (int u = 0; u < BT_D; ++u) {
        const float* wr = s + u * BT_D;
        float acc = s[2 * BT_D * BT_D + u];
#pragma unroll
        for (int j = 0; j < BT_D; ++j) acc = fmaf(wr[j], z.v[j], acc);
        a.v[u] = acc;
        // keep the 576 weight reads of a layer from being hoisted in front of its fma chains (left alone the backward
        // kernel takes 256 VGPRs and 3 KB of scratch per lane: 118 us for 60k points instead of ~15)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// out = ff.2(relu(ff.0(norm2(x1))))   (example/transformer.py:162 without the residual and the dropout around it)
__global__ __launch_bounds__(LN_THREADS) void ln_ffn_fwd_kernel(const float* __restrict__ x1, FfnW p, int N,
                                                                float* __restrict__ out) {
    __shared__ float s[FFN_W];
    stage_ffn(p, s);
    const int n = blockIdx.x * LN_THREADS + threadIdx.x;
    if (n >= N) return;
    Row xr, xhat, z, a, o;
    xr.load(x1 + (size_t)n * BT_D);
    float rstd;
    ln_stats(xr, p.eps, xhat, rstd);
    ffn_hidden(xhat, s, z, a);
#pragma unroll
    for (int i = 0; i < BT_D; ++i) {
        const float* wr = s + BT_D * BT_D + i * BT_D;
        float acc = s[2 * BT_D * BT_D + BT_D + i];
#pragma unroll
        for (int u = 0; u < BT_D; ++u) acc = fmaf(wr[u], fmaxf(a.v[u], 0.f), acc);
        o.v[i] = acc;
        __builtin_amdgcn_sched_barrier(0);
    }
    o.store(out + (size_t)n * BT_D);
}

// backward of the above for the upstream gradient d_out: d x1 (this branch's share), and the rows the two weight
// gradients need -- z = norm2(x1), h = relu(ff.0(z)), dh = d(ff.0 output) -- plus the LayerNorm parameter partials
__global__ __launch_bounds__(LN_THREADS) void ln_ffn_bwd_kernel(const float* __restrict__ x1,
                                                                const float* __restrict__ d_out, FfnW p, int N,
                                                                float* __restrict__ dx1, float* __restrict__ z_out,
                                                                float* __restrict__ h_out, float* __restrict__ dh_out,
                                                                float* __restrict__ partial, float* __restrict__ dump) {
    __shared__ float s[FFN_W];
    __shared__ float wave_s[(LN_THREADS / 64) * 2 * BT_D];
    stage_ffn(p, s);
    const int n = blockIdx.x * LN_THREADS + threadIdx.x;
    const bool valid = n < N;
    const size_t row = (size_t)(valid ? n : N - 1) * BT_D;
    // (phase by phase, every row stored as soon as it is final, so that at most ~four rows are live at a time: with all
    //  eight rows and the hoisted weight reads live the kernel took 256 VGPRs and 3 KB of scratch per lane)
    Row xhat, dz;
    float rstd;
    unsigned int relu_mask = 0;   // bit u: ff.0 output u is positive
    {
        Row xr, z, a;
        xr.load(x1 + row);
        ln_stats(xr, p.eps, xhat, rstd);
        ffn_hidden(xhat, s, z, a);
        z.store(valid ? z_out + row : dump);
#pragma unroll
        for (int u = 0; u < BT_D; ++u) {
            relu_mask |= a.v[u] > 0.f ? 1u << u : 0u;
            a.v[u] = fmaxf(a.v[u], 0.f);
        }
        a.store(valid ? h_out + row : dump);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        Row go, dh;
        go.load(d_out + row);
        // dh = (W2^T d_out) * [a > 0]
#pragma unroll
        for (int u = 0; u < BT_D; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < BT_D; ++i) acc = fmaf(s[BT_D * BT_D + i * BT_D + u], go.v[i], acc);
            dh.v[u] = (relu_mask >> u) & 1u ? acc : 0.f;
            __builtin_amdgcn_sched_barrier(0);
        }
        dh.store(valid ? dh_out + row : dump);
        // dz = W1^T dh
#pragma unroll
        for (int j = 0; j < BT_D; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < BT_D; ++u) acc = fmaf(s[u * BT_D + j], dh.v[u], acc);
            dz.v[j] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    {
        Row dx;
        ln_backward(dz, xhat, rstd, s + 2 * BT_D * BT_D + 2 * BT_D, dx);
        dx.store(valid ? dx1 + row : dump);
    }
    ln_param_partials<LN_THREADS>(dz, xhat, valid, wave_s, partial);
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int n_wgs_points(int N) { return (N + BT_POINTS - 1) / BT_POINTS; }
inline int n_wgs_rows(int N) { return (N + LN_THREADS - 1) / LN_THREADS; }

int wgrad_launch(const float* dY, const float* X, int N, int O, float* d_weight, float* d_bias, float* partial,
                 hipStream_t st) {
    const int wgs = n_wgs_points(N);
    if (O > 32) {
        constexpr int COLS = 192, GROUPS = 4;
        hipLaunchKernelGGL((wgrad_kernel<COLS, GROUPS>), dim3(wgs, (O + COLS - 1) / COLS), dim3(COLS * GROUPS), 0, st, dY, X,
                           N, O, partial);
    } else {
        constexpr int COLS = 32, GROUPS = 16;
        hipLaunchKernelGGL((wgrad_kernel<COLS, GROUPS>), dim3(wgs, 1), dim3(COLS * GROUPS), 0, st, dY, X, N, O, partial);
    }
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3(((BT_D + 1) * O + HEPT_FSUM_OUT - 1) / HEPT_FSUM_OUT), dim3(256), 0, st,
                       partial, wgs, O, d_weight, d_bias);
    return hept_launch_status();
}

}  // namespace

extern "C" size_t hept_rows_wgrad_scratch_bytes(int N, int O) {
    return N < 1 || O < 1 ? 0 : al256((size_t)n_wgs_points(N) * (BT_D + 1) * O * sizeof(float));
}

// d_weight (O, 24) = dY^T . X, d_bias (O) = column sums of dY (or NULL): the weight gradient of a Linear(24 -> O) whose
// input rows were X (N, 24) and whose output gradient is dY (N, O); replaces torch's dY.t() @ X.
extern "C" int hept_rows_wgrad(const float* dY, const float* X, int N, int O, int J, float* d_weight, float* d_bias,
                               void* scratch, size_t scratch_bytes, void* stream) {
    if (!dY || !X || !d_weight || !scratch) return HEPT_ERR_ARG;
    if (N < 1 || O < 1 || J != BT_D) return HEPT_ERR_SHAPE;
    if (scratch_bytes < hept_rows_wgrad_scratch_bytes(N, O)) return HEPT_ERR_ARG;
    return wgrad_launch(dY, X, N, O, d_weight, d_bias, static_cast<float*>(scratch), (hipStream_t)stream);
}

extern "C" size_t hept_ln_scratch_bytes(int N) {   // LayerNorm parameter partials + one dump row
    return N < 1 ? 0 : al256((size_t)n_wgs_rows(N) * 2 * BT_D * sizeof(float)) + 256;
}
static inline float* ln_dump_row(float* partial, int N) {
    return partial + al256((size_t)n_wgs_rows(N) * 2 * BT_D * sizeof(float)) / sizeof(float);
}

// LayerNorm(24) backward (example/transformer.py:155 under autograd): dx (N, 24), the normalised rows xn (N, 24),
// d_ln_w, d_ln_b (24 each) from x and dxn = d LayerNorm(x)
extern "C" int hept_ln_bwd(const float* x, const float* dxn, const float* ln_w, const float* ln_b, float eps, int N,
                           int D, float* dx, float* xn, float* d_ln_w, float* d_ln_b, void* scratch,
                           size_t scratch_bytes, void* stream) {
    if (!x || !dxn || !ln_w || !ln_b || !dx || !xn || !d_ln_w || !d_ln_b || !scratch) return HEPT_ERR_ARG;
    if (N < 1 || D != BT_D) return HEPT_ERR_SHAPE;
    if (scratch_bytes < hept_ln_scratch_bytes(N)) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* partial = static_cast<float*>(scratch);
    const int wgs = n_wgs_rows(N);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(wgs), dim3(LN_THREADS), 0, st, x, dxn, ln_w, ln_b, eps, N, dx, xn, partial,
                       ln_dump_row(partial, N));
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * BT_D + HEPT_FSUM_OUT - 1) / HEPT_FSUM_OUT), dim3(256), 0, st, partial,
                       wgs, 2 * BT_D, d_ln_w, d_ln_b, BT_D);
    return hept_launch_status();
}

// out (N, 24) = ff.2(relu(ff.0(norm2(x1)))), example/transformer.py:162 (the residual and the dropouts stay outside)
extern "C" int hept_ln_ffn_fwd(const float* x1, const float* ln_w, const float* ln_b, float eps, const float* w1,
                               const float* b1, const float* w2, const float* b2, int N, int D, float* out,
                               void* stream) {
    if (!x1 || !ln_w || !ln_b || !w1 || !b1 || !w2 || !b2 || !out) return HEPT_ERR_ARG;
    if (N < 1 || D != BT_D) return HEPT_ERR_SHAPE;
    const FfnW p{ln_w, ln_b, w1, b1, w2, b2, eps};
    hipLaunchKernelGGL(ln_ffn_fwd_kernel, dim3(n_wgs_rows(N)), dim3(LN_THREADS), 0, (hipStream_t)stream, x1, p, N, out);
    return hept_launch_status();
}

extern "C" size_t hept_ln_ffn_bwd_scratch_bytes(int N) {
    if (N < 1) return 0;
    return 3 * al256((size_t)N * BT_D * sizeof(float)) + hept_ln_scratch_bytes(N) + hept_rows_wgrad_scratch_bytes(N, BT_D);
}

// backward of hept_ln_ffn_fwd: d_x1 (N, 24) and the gradients of the eight parameters, from x1 and d_out
extern "C" int hept_ln_ffn_bwd(const float* x1, const float* d_out, const float* ln_w, const float* ln_b, float eps,
                               const float* w1, const float* b1, const float* w2, const float* b2, int N, int D,
                               float* d_x1, float* d_ln_w, float* d_ln_b, float* d_w1, float* d_b1, float* d_w2,
                               float* d_b2, void* scratch, size_t scratch_bytes, void* stream) {
    if (!x1 || !d_out || !ln_w || !ln_b || !w1 || !b1 || !w2 || !b2 || !d_x1 || !d_ln_w || !d_ln_b || !d_w1 || !d_b1 ||
        !d_w2 || !d_b2 || !scratch)
        return HEPT_ERR_ARG;
    if (N < 1 || D != BT_D) return HEPT_ERR_SHAPE;
    if (scratch_bytes < hept_ln_ffn_bwd_scratch_bytes(N)) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* ws = static_cast<char*>(scratch);
    const size_t rows = al256((size_t)N * BT_D * sizeof(float));
    float* z = reinterpret_cast<float*>(ws);
    float* h = reinterpret_cast<float*>(ws + rows);
    float* dh = reinterpret_cast<float*>(ws + 2 * rows);
    float* ln_part = reinterpret_cast<float*>(ws + 3 * rows);
    float* wg_part = reinterpret_cast<float*>(ws + 3 * rows + hept_ln_scratch_bytes(N));
    const FfnW p{ln_w, ln_b, w1, b1, w2, b2, eps};
    const int wgs = n_wgs_rows(N);
    hipLaunchKernelGGL(ln_ffn_bwd_kernel, dim3(wgs), dim3(LN_THREADS), 0, st, x1, d_out, p, N, d_x1, z, h, dh, ln_part,
                       ln_dump_row(ln_part, N));
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * BT_D + HEPT_FSUM_OUT - 1) / HEPT_FSUM_OUT), dim3(256), 0, st, ln_part,
                       wgs, 2 * BT_D, d_ln_w, d_ln_b, BT_D);
    int rc = wgrad_launch(d_out, h, N, BT_D, d_w2, d_b2, wg_part, st);   // ff.2: out = W2 h + b2
    if (rc) return rc;
    return wgrad_launch(dh, z, N, BT_D, d_w1, d_b1, wg_part, st);         // ff.0: a = W1 z + b1
}
