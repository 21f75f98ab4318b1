// Micro-benchmark: what does the partial-row read pattern of combine_out cost against a fully coalesced read of the
// same bytes?  part = (T, N, H, 64 B) packed rows.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/row_read_patterns.hip -o /tmp/rrp && /tmp/rrp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int N = 60032, H = 8, T = 3;

// A: lane (li, hh) of a wave owns point li, heads 2j + hh: 4 x 16 B of one 64-B row per (table, head pair)
__global__ __launch_bounds__(256) void pattern_rows(const char* __restrict__ part, unsigned int* __restrict__ sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, hh = lane >> 5;
    const int tiles = (N + 31) / 32;
    unsigned int acc = 0;
    for (int tile = blockIdx.x * 4 + w; tile < tiles; tile += gridDim.x * 4) {
        const int n = min(tile * 32 + li, N - 1);
#pragma unroll
        for (int hp = 0; hp < H; hp += 2) {
            u32x4 v[T][4];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const u32x4* src = reinterpret_cast<const u32x4*>(part + (((size_t)t * N + n) * H + hp + hh) * 64);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[t][q] = src[q];
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += v[t][q][0] ^ v[t][q][3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// B: the same bytes per wave tile, but every load instruction covers 8 full 128-B lines (8 lanes per line)
__global__ __launch_bounds__(256) void pattern_lines(const char* __restrict__ part, unsigned int* __restrict__ sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tiles = (N + 31) / 32;
    unsigned int acc = 0;
    for (int tile = blockIdx.x * 4 + w; tile < tiles; tile += gridDim.x * 4) {
#pragma unroll
        for (int hp = 0; hp < H; hp += 2) {
            u32x4 v[T][4];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = k * 64 + lane, pt = c >> 3, piece = c & 7;
                    const int n = min(tile * 32 + pt, N - 1);
                    v[t][k] = *reinterpret_cast<const u32x4*>(part + (((size_t)t * N + n) * H + hp) * 64 + piece * 16);
                }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += v[t][q][0] ^ v[t][q][3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// C: plain streaming read of the whole buffer
__global__ __launch_bounds__(256) void pattern_stream(const char* __restrict__ part, size_t n16, unsigned int* __restrict__ sink) {
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 v = reinterpret_cast<const u32x4*>(part)[i];
        acc += v[0] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)T * N * H * 64;
    char* part;
    unsigned int* sink;
    hipMalloc(&part, bytes);
    hipMalloc(&sink, 4);
    hipMemset(part, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {469, 938, 1876}) {
        for (int which = 0; which < 3; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(pattern_rows, dim3(grid), dim3(256), 0, 0, part, sink);
                if (which == 1) hipLaunchKernelGGL(pattern_lines, dim3(grid), dim3(256), 0, 0, part, sink);
                if (which == 2) hipLaunchKernelGGL(pattern_stream, dim3(grid * 2), dim3(256), 0, 0, part, bytes / 16, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("grid %4d  %-14s %7.2f us  %6.2f TB/s\n", grid, which == 0 ? "rows(current)" : which == 1 ? "full lines" : "stream",
                   best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
