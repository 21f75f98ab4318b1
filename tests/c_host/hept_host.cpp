// A PyTorch-free host for the C ABI (include/hept_hip.h): reads one problem from a binary file, runs hept_forward on
// the GPU with plain hipMalloc'd buffers on its own stream, writes the (N, D) output back.  Built and driven by
// tests/test_gpu_c_host.py; shows what a C/C++ (or any FFI) caller of libhept_hip.so looks like.
//
// File layout (little endian): int32 header[9] = {N, H, D, C, K, T, B, precision, 0}, then f32 arrays
//   q, k, v (N*H*D each), coords (N*C), w_rpe (H*D * (C-1)*K), alpha (H*(D+C)*T), out_w (D * H*D), out_b (D),
//   then int64 codes (T*H*N).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/hept_hip.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                              \
        }                                                                          \
    } while (0)

template <typename T>
static bool read_to_device(std::FILE* f, size_t count, T** dev) {
    std::vector<T> host(count);
    if (std::fread(host.data(), sizeof(T), count, f) != count) return false;
    if (hipMalloc(reinterpret_cast<void**>(dev), count * sizeof(T)) != hipSuccess) return false;
    return hipMemcpy(*dev, host.data(), count * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]);
        return 1;
    }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hd[9];
    if (std::fread(hd, sizeof(int32_t), 9, f) != 9) return 1;
    const int N = hd[0], H = hd[1], D = hd[2], C = hd[3], K = hd[4], T = hd[5], B = hd[6], precision = hd[7];
    if (hept_abi_version() < 7) return 3;
    if (hept_check_shape(N, H, D, C, T, B) != HEPT_OK) {
        std::fprintf(stderr, "unsupported shape\n");
        return 3;
    }
    const size_t nhd = (size_t)N * H * D;
    float *q, *k, *v, *coords, *w_rpe, *alpha, *out_w, *out_b, *out;
    int64_t* codes;
    if (!read_to_device(f, nhd, &q) || !read_to_device(f, nhd, &k) || !read_to_device(f, nhd, &v) ||
        !read_to_device(f, (size_t)N * C, &coords) || !read_to_device(f, (size_t)H * D * (C - 1) * K, &w_rpe) ||
        !read_to_device(f, (size_t)H * (D + C) * T, &alpha) || !read_to_device(f, (size_t)D * H * D, &out_w) ||
        !read_to_device(f, (size_t)D, &out_b) || !read_to_device(f, (size_t)T * H * N, &codes)) {
        std::fprintf(stderr, "short problem file or allocation failure\n");
        return 2;
    }
    std::fclose(f);
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const size_t ws_bytes = hept_workspace_bytes(N, H, D, C, T, B, precision);
    void* ws;
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&out), (size_t)N * D * sizeof(float)));
    int rc = 0;
    for (int rep = 0; rep < 2 && rc == 0; ++rep)  // twice: the call is re-entrant on the same workspace
        rc = hept_forward(q, k, v, coords, codes, w_rpe, alpha, out_w, out_b, N, H, D, C, K, T, B, precision, ws, ws_bytes,
                          out, stream);
    if (rc != HEPT_OK) {
        std::fprintf(stderr, "hept_forward failed: %d\n", rc);
        return 4;
    }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> host((size_t)N * D);
    HIP_OK(hipMemcpy(host.data(), out, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::FILE* g = std::fopen(argv[2], "wb");
    if (!g || std::fwrite(host.data(), sizeof(float), host.size(), g) != host.size()) return 1;
    std::fclose(g);
    return 0;
}
