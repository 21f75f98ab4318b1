// combine_out: OR-combine over tables + output projection; reduce_tables for table sharding.
//
// Replaces (reference file:line):
//   out = o.sum(0) / logits.sum(0)                         example/hept.py:79
//   out_linear(rearrange(out, "h n d -> n (h d)"))         example/hept.py:80
//
// HBM-bound: reads Tl*H rows of 128 B per point (part layout (Tl, N, H, 32): the H rows of a
// point are one contiguous 1-KiB run per table), writes D floats per point.  One wave owns 32
// points: lane (point = lane & 31, half = lane >> 5) loads 64 contiguous bytes of each
// (table, head) row with 16-B loads, sums the tables in the reference's order, divides by the
// denominator column and feeds the 32 x (H*32) tile straight to v_mfma_f32_32x32x2_f32 against
// the LDS-resident, zero-padded transpose of out_linear.weight (exact fp32 fma chain).
#include "common.h"

namespace {

constexpr int CMB_THREADS = 256;
constexpr int CMB_WAVES = CMB_THREADS / HEPT_WAVE;

__global__ __launch_bounds__(CMB_THREADS) void combine_out_kernel(const float* __restrict__ part, int Tl, int N,
                                                                  int H, int D, int n0, int n_count,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wt_s[];  // [H][32 (d, zero padded)][32 (c, zero padded)]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int HD = H * D;
    for (int i = tid; i < H * 32 * 32; i += CMB_THREADS) {
        const int c = i & 31, d = (i >> 5) & 31, h = i >> 10;
        wt_s[i] = (c < D && d < D) ? W[(size_t)c * HD + h * D + d] : 0.f;  // W is (D, H*D) row-major
    }
    const float bia = (li < D && bias) ? bias[li] : 0.f;
    __syncthreads();
    const size_t tstride = (size_t)N * H * 32;

    const int n_tiles = (n_count + 31) / 32;
    for (int tile = blockIdx.x * CMB_WAVES + w; tile < n_tiles; tile += gridDim.x * CMB_WAVES) {
        const int i = tile * 32 + li;
        const int n = n0 + (i < n_count ? i : n_count - 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* row0 = part + (size_t)n * H * 32 + 16 * hh;
        for (int h = 0; h < H; ++h) {
            const f32x4* src = reinterpret_cast<const f32x4*>(row0 + h * 32);
            f32x4 s[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) s[c4] = src[c4];
            for (int t = 1; t < Tl; ++t) {
                const f32x4* st = reinterpret_cast<const f32x4*>(row0 + h * 32 + (size_t)t * tstride);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) s[c4] += st[c4];
            }
            // denominator = column D of the row: held by the lane half that owns columns [16*hh, 16*hh+16)
            float den = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (16 * hh + 4 * c4 + u == D) den = s[c4][u];
            const int owner = li + 32 * (D >> 4);
            den = __shfl(den, owner);
            const float* wrow = wt_s + (size_t)h * 1024 + (16 * hh) * 32 + li;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[c4][u] / den, wrow[(4 * c4 + u) * 32], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i2 = tile * 32 + hept_acc_row(r, hh);
            if (i2 < n_count && li < D) out[(size_t)i2 * D + li] = acc[r] + bia;
        }
    }
}

__global__ __launch_bounds__(256) void reduce_tables_kernel(const f32x4* __restrict__ part, int Tl, size_t n4,
                                                            f32x4* __restrict__ acc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = part[i];
        for (int t = 1; t < Tl; ++t) s += part[i + (size_t)t * n4];
        acc[i] = s;
    }
}

}  // namespace

extern "C" int hept_reduce_tables(const float* part, int Tl, int N, int H, float* acc, void* stream) {
    if (!part || !acc) return HEPT_ERR_ARG;
    if (Tl < 1 || N < 1 || H < 1) return HEPT_ERR_SHAPE;
    const size_t n4 = (size_t)N * H * 32 / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_tables_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(part), Tl, n4, reinterpret_cast<f32x4*>(acc));
    return hept_launch_status();
}

extern "C" int hept_combine_out(const float* part, int Tl, int N, int H, int D, int n0, int n_count,
                                const float* out_weight, const float* out_bias, float* out, void* stream) {
    if (!part || !out_weight || !out) return HEPT_ERR_ARG;
    if (Tl < 1 || N < 1 || H < 1 || H > 15 || D < 1 || D > 28 || n0 < 0 || n_count < 0 || n0 + n_count > N)
        return HEPT_ERR_SHAPE;
    if (n_count == 0) return HEPT_OK;
    const size_t lds = sizeof(float) * (size_t)H * 32 * 32;
    const int n_tiles = (n_count + 31) / 32;
    const int wgs = (n_tiles + CMB_WAVES - 1) / CMB_WAVES;
    const int grid = wgs < 2048 ? wgs : 2048;
    hipLaunchKernelGGL(combine_out_kernel, dim3(grid), dim3(CMB_THREADS), lds, (hipStream_t)stream, part, Tl, N, H, D,
                       n0, n_count, out_weight, out_bias, out);
    return hept_launch_status();
}
