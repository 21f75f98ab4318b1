// combine_out: OR-combine over tables + output projection; reduce_tables for table sharding.
//
// Replaces (reference file:line):
//   out = o.sum(0) / logits.sum(0)                         example/hept.py:79
//   out_linear(rearrange(out, "h n d -> n (h d)"))         example/hept.py:80
//
// HBM-bound: reads Tl*H rows of 128 B per point (part layout (Tl, N, H, 32): the H rows of a
// point are one contiguous 1-KiB run per table), writes D floats per point.
#include "common.h"

namespace {

constexpr int CMB_THREADS = 256;
constexpr int CMB_POINTS = CMB_THREADS / 32;  // one 32-lane half-wave per point

__global__ __launch_bounds__(CMB_THREADS) void combine_out_kernel(const float* __restrict__ part, int Tl, int N,
                                                                  int H, int D, int n0, int n_count,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HD = H * D;
    float* wt_s = smem;           // [HD][D] : W transposed, lane c reads consecutive addresses
    float* o_s = smem + HD * D;   // [CMB_POINTS][HD]
    const int tid = threadIdx.x, grp = tid >> 5, l = tid & 31;
    for (int i = tid; i < HD * D; i += CMB_THREADS) {
        const int c = i / HD, j = i - c * HD;  // W is (D, HD) row-major
        wt_s[j * D + c] = W[i];
    }
    const float bia = (l < D && bias) ? bias[l] : 0.f;
    const size_t tstride = (size_t)N * H * 32;

    const int n_groups = (n_count + CMB_POINTS - 1) / CMB_POINTS;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int i = g * CMB_POINTS + grp;
        const bool live = i < n_count;
        const int n = n0 + (live ? i : 0);
        __syncthreads();  // o_s of the previous group fully consumed (and wt_s ready on entry)
        for (int h = 0; h < H; ++h) {
            const float* src = part + ((size_t)n * H + h) * 32 + l;
            float s = src[0];
            for (int t = 1; t < Tl; ++t) s += src[(size_t)t * tstride];
            const float den = __shfl(s, D, 32);
            if (l < D) o_s[grp * HD + h * D + l] = s / den;
        }
        __syncthreads();
        if (live && l < D) {
            float acc = bia;
            const float* o = o_s + grp * HD;
            for (int j = 0; j < HD; ++j) acc = fmaf(o[j], wt_s[j * D + l], acc);
            out[(size_t)i * D + l] = acc;
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
    if (Tl < 1 || N < 1 || H < 1 || D < 1 || D > 28 || n0 < 0 || n_count < 0 || n0 + n_count > N)
        return HEPT_ERR_SHAPE;
    if (n_count == 0) return HEPT_OK;
    const size_t lds = sizeof(float) * ((size_t)H * D * D + (size_t)CMB_POINTS * H * D);
    if (lds > 65536) return HEPT_ERR_SHAPE;
    const int n_groups = (n_count + CMB_POINTS - 1) / CMB_POINTS;
    const int grid = n_groups < 2048 ? n_groups : 2048;
    hipLaunchKernelGGL(combine_out_kernel, dim3(grid), dim3(CMB_THREADS), lds, (hipStream_t)stream, part, Tl, N, H, D,
                       n0, n_count, out_weight, out_bias, out);
    return hept_launch_status();
}
