// prepare_input on the GPU: quantile regions -> packed AND code -> pad to block multiples.
//
// Replaces, for one batch of clouds (reference file:line):
//   argsort(eta), argsort(phi) per cloud + quantile_partition   example/transformer.py:44-51, example/hept_utils.py:6-14
//   bit_shift x2                                                example/transformer.py:10-13,55-56
//   pad_and_unpad + the three gathers by pad_seq                example/transformer.py:16-32,59-62
//
// Integer/byte work on a few 1e5 elements: launch-latency bound, not bandwidth bound.  All sorts go through
// hept_segmented_argsort (stable, exact); region ids use integer division (rank and width are exact
// integers in fp32, so `rank // width` of the reference is the integer quotient); the bit counts
// ceil(log2(max + 1)) are bit lengths.  The reference ranks with an unstable argsort: ties between equal
// coordinates / equal codes are broken by index here (as in the torch mirror hept_amd/prep.py).
#include "common.h"

namespace {

constexpr int PT = 256;

__device__ __forceinline__ int bit_length(int m) { return m <= 0 ? 0 : 32 - __clz(m); }

// keys[(c*2 + a)][j] = coordinate a of the j-th point of cloud c (row pitch L, only the first n_c entries are
// written and sorted: seg_len[c*2 + a] = n_c)
__global__ __launch_bounds__(PT) void coord_keys_kernel(const float* __restrict__ coords, int C,
                                                        const int* __restrict__ cloud_start, int L,
                                                        float* __restrict__ keys, int* __restrict__ seg_len) {
    const int seg = blockIdx.y, c = seg >> 1, a = seg & 1;
    const int j = blockIdx.x * PT + threadIdx.x;
    const int s = cloud_start[c], n_c = cloud_start[c + 1] - s;
    if (j == 0) seg_len[seg] = n_c;
    if (j < n_c) keys[(size_t)seg * L + j] = coords[(size_t)(s + j) * C + a];
}

__device__ __forceinline__ void row_bits_body(const int* __restrict__ cloud_start, int n_clouds,
                                              const float* __restrict__ regions, int T, int H,
                                              int* __restrict__ row_max, int row, int lane);

// rank[a][point] = position of the point in its cloud's ascending order of coordinate a.  The launch also carries, as
// the blocks of one more grid row (blockIdx.y == S), the row maxima of the region ids (row_bits_body below: a few
// dozen waves that depend on the cloud sizes only -- a launch of their own cost ~7 us of this launch-bound call)
__global__ __launch_bounds__(PT) void rank_scatter_kernel(const int* __restrict__ pos,
                                                          const int* __restrict__ cloud_start, int L, int n_raw,
                                                          int* __restrict__ rank, int S, int n_clouds,
                                                          const float* __restrict__ regions, int T, int H,
                                                          int* __restrict__ row_max) {
    if ((int)blockIdx.y == S) {
        for (int row = blockIdx.x * (PT / 64) + (threadIdx.x >> 6); row < T * H; row += gridDim.x * (PT / 64))
            row_bits_body(cloud_start, n_clouds, regions, T, H, row_max, row, threadIdx.x & 63);
        return;
    }
    const int seg = blockIdx.y, c = seg >> 1, a = seg & 1;
    const int r = blockIdx.x * PT + threadIdx.x;
    const int s = cloud_start[c], n_c = cloud_start[c + 1] - s;
    if (r < n_c) rank[(size_t)a * n_raw + s + pos[(size_t)seg * L + r]] = r;
}

// region id of a rank: rank // ceil(n_c / regions) + 1, with the reference's fp32 evaluation of the width
// (`n / regions` with a Python-int numerator is reciprocal-times-n in torch, hept_amd/prep.py)
__device__ __forceinline__ int region_of(int rank, int n_c, float regions) {
    const float width = ceilf((1.0f / regions) * (float)n_c);
    return rank / (int)width + 1;
}

// cloud of a raw point: binary search in cloud_start (points of a cloud are contiguous)
__device__ __forceinline__ int cloud_of(const int* __restrict__ cloud_start, int n_clouds, int n) {
    int lo = 0, hi = n_clouds;  // invariant: cloud_start[lo] <= n < cloud_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cloud_start[mid] <= n) lo = mid; else hi = mid;
    }
    return lo;
}

// Sort key of the table-0 / head-0 codes.  A bare float(code) would hand the sort ~150 distinct values for 60 000
// points: tie groups of 400, which its in-group ranking pays for quadratically (60 us for this one segment).  The
// stable order wanted is (code, index), so the key carries the leading index bits below the code,
//     key = code << k | index >> s,      k = 24 - bits(code),  s = bits(index) - k   (both >= 0),
// exact in fp32 (< 2^24), monotone in (code, index): equal keys are now runs of at most 2^s consecutive indices and
// the final order is unchanged.  Every key lies in [0, 2^24): the sort is told so and skips its range pass.
__device__ __forceinline__ float code_sort_key(int64_t code, int n, int n_raw, int n_clouds, int packed_max_row0) {
    const int code_bits = bit_length(n_clouds - 1) + bit_length(packed_max_row0);   // codes < 2^24 (checked by the caller)
    const int k = code_bits < 24 ? 24 - code_bits : 0;
    const int idx_bits = bit_length(n_raw - 1);
    const int sh = idx_bits > k ? idx_bits - k : 0;
    return (float)((code << k) | (int64_t)(n >> sh));
}

// final codes of one (table, head) row: (cloud << bits1) | (phi_region << bits0) | eta_region, the bit widths from
// row_bits_kernel
__global__ __launch_bounds__(PT) void codes_kernel(const int* __restrict__ rank, const int* __restrict__ cloud_start,
                                                   int n_clouds, int n_raw, const float* __restrict__ regions, int T,
                                                   int H, const int* __restrict__ row_max /* [2][T*H] */,
                                                   int64_t* __restrict__ codes_raw, float* __restrict__ code_keys) {
    const int row = blockIdx.y, t = row / H, h = row % H;
    const int n = blockIdx.x * PT + threadIdx.x;
    if (n >= n_raw) return;
    const int c = cloud_of(cloud_start, n_clouds, n);
    const int n_c = cloud_start[c + 1] - cloud_start[c];
    const int eta = region_of(rank[n], n_c, regions[((size_t)t * 2 + 0) * H + h]);
    const int phi = region_of(rank[(size_t)n_raw + n], n_c, regions[((size_t)t * 2 + 1) * H + h]);
    const int p1 = (phi << bit_length(row_max[row])) | eta;
    const int64_t code = ((int64_t)c << bit_length(row_max[(size_t)T * H + row])) | (int64_t)p1;
    codes_raw[(size_t)row * n_raw + n] = code;
    if (row == 0) code_keys[n] = code_sort_key(code, n, n_raw, n_clouds, row_max[(size_t)T * H]);
}

// Largest region ids of a row without touching the points: ranks run 0 .. n_c - 1 in every cloud and region_of is
// monotone in the rank, so the row maximum is the maximum over the clouds of region_of(n_c - 1).  The packing only
// needs bit lengths: bit_length((phi << b) | eta) = bit_length(phi) + b for phi >= 1 (region ids start at 1), so
// row_max[1][row] = (phi_max << b) | eta_max has the bit length of the true maximum of the packed values.
__device__ __forceinline__ void row_bits_body(const int* __restrict__ cloud_start, int n_clouds,
                                              const float* __restrict__ regions, int T, int H,
                                              int* __restrict__ row_max /* [2][T*H] */, int row, int lane) {
    const int t = row / H, h = row % H;
    const float r_eta = regions[((size_t)t * 2 + 0) * H + h], r_phi = regions[((size_t)t * 2 + 1) * H + h];
    int eta = 0, phi = 0;
    for (int c = lane; c < n_clouds; c += 64) {
        const int n_c = cloud_start[c + 1] - cloud_start[c];
        if (n_c > 0) {
            eta = max(eta, region_of(n_c - 1, n_c, r_eta));
            phi = max(phi, region_of(n_c - 1, n_c, r_phi));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        eta = max(eta, __shfl_xor(eta, off));
        phi = max(phi, __shfl_xor(phi, off));
    }
    if (lane == 0) {
        row_max[row] = eta;
        row_max[(size_t)T * H + row] = (phi << bit_length(eta)) | eta;
    }
}

// one thread per padded slot: gather index, un-pad mask, padded coords; blockIdx.y > 0: padded codes of one row
__global__ __launch_bounds__(PT) void pad_gather_kernel(const int* __restrict__ cloud_start,
                                                        const int* __restrict__ pad_start, int n_clouds, int n_raw,
                                                        int n_pad, int B, const int* __restrict__ by_code,
                                                        const float* __restrict__ coords, int C,
                                                        const int64_t* __restrict__ codes_raw, int rows,
                                                        int64_t* __restrict__ pad_seq, unsigned char* __restrict__ unpad,
                                                        float* __restrict__ coords_pad, int64_t* __restrict__ codes_pad) {
    const int s = blockIdx.x * PT + threadIdx.x;
    if (s >= n_pad) return;
    const int c = cloud_of(pad_start, n_clouds, s);
    const int j = s - pad_start[c];
    const int n_c = cloud_start[c + 1] - cloud_start[c];
    const bool real = j < n_c;
    int src;
    if (real) {
        src = cloud_start[c] + j;
    } else {
        // pad slot j - n_c copies the point at sorted position raw_end - B + (j - n_c) of the table-0/head-0
        // code order; a cloud smaller than B reaches into the previous cloud, a negative index wraps
        // (torch indexing), exactly as the reference does
        int sp = cloud_start[c + 1] - B + (j - n_c);
        if (sp < 0) sp += n_raw;
        src = by_code[sp];
    }
    const int part = blockIdx.y;
    if (part == 0) {
        pad_seq[s] = src;
        unpad[s] = real ? 1 : 0;
        for (int a = 0; a < C; ++a) coords_pad[(size_t)s * C + a] = coords[(size_t)src * C + a];
    } else {
        const int row = part - 1;
        codes_pad[(size_t)row * n_pad + s] = codes_raw[(size_t)row * n_raw + src];
    }
}

// ---- the src variant's caller side (src/models/baselines/transformer.py:43-57): ONE cloud, padded once at its end
// keys[a][n] = coordinate a of point n, +inf for the padding slots (they sort last, in index order); the padded
// coordinate rows (zero rows for the padding, :57) and the padded feature rows (zero rows, :46) are written here too
__global__ __launch_bounds__(PT) void src_pad_keys_kernel(const float* __restrict__ x, int F,
                                                          const float* __restrict__ coords, int C, int raw_size, int N,
                                                          float* __restrict__ keys, float* __restrict__ x_pad,
                                                          float* __restrict__ coords_pad) {
    const size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    const int part = blockIdx.y;
    if (part == 0) {
        if (i < (size_t)N) {
            const bool real = i < (size_t)raw_size;
            keys[i] = real ? coords[i * C] : INFINITY;
            keys[(size_t)N + i] = real ? coords[i * C + 1] : INFINITY;
            for (int a = 0; a < C; ++a) coords_pad[i * C + a] = real ? coords[i * C + a] : 0.f;
        }
    } else if (x_pad) {
        const size_t total = (size_t)N * F, j = ((size_t)(part - 1) * gridDim.x + blockIdx.x) * PT + threadIdx.x;
        const size_t step = (size_t)(gridDim.y - 1) * gridDim.x * PT;
        for (size_t e = j; e < total; e += step) x_pad[e] = e < (size_t)raw_size * F ? x[e] : 0.f;
    }
}

// rank[a][point] = position of the point in the ascending order of coordinate a
__global__ __launch_bounds__(PT) void src_rank_kernel(const int* __restrict__ pos, int N, int* __restrict__ rank) {
    const int a = blockIdx.y, r = blockIdx.x * PT + threadIdx.x;
    if (r < N) rank[(size_t)a * N + pos[(size_t)a * N + r]] = r;
}

// region_indices[a][(t, h)][n] = rank // ceil(N / regions[t][a][h]) + 1 as float (quantile_partition,
// src/models/model_utils/hash_utils.py:14-22, with torch's reciprocal-times-n evaluation of `int / tensor`)
__global__ __launch_bounds__(PT) void src_region_kernel(const int* __restrict__ rank, int N,
                                                        const float* __restrict__ regions, int T, int H,
                                                        float* __restrict__ eta, float* __restrict__ phi) {
    const int row = blockIdx.y % (T * H), a = blockIdx.y / (T * H), t = row / H, h = row % H;
    const int n = blockIdx.x * PT + threadIdx.x;
    if (n >= N) return;
    const int id = region_of(rank[(size_t)a * N + n], N, regions[((size_t)t * 2 + a) * H + h]);
    (a ? phi : eta)[(size_t)row * N + n] = (float)id;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t hept_prepare_src_workspace_bytes(int N) {
    if (N < 1) return 0;
    return al256((size_t)2 * N * 4) * 3 + al256(hept_argsort_workspace_bytes(2, N));
}

// x (raw_size, F) f32 or NULL (then x_pad is not written); coords (raw_size, C); regions (T, 2, H).
// Outputs: x_pad (N, F), coords_pad (N, C), eta / phi (T*H, N) f32 = kwargs["region_indices"]; N = raw_size rounded up
// to a multiple of the block size by the caller.
extern "C" int hept_prepare_input_src(const float* x, int F, const float* coords, int C, int raw_size, int N,
                                      const float* regions, int T, int H, void* workspace, size_t workspace_bytes,
                                      float* x_pad, float* coords_pad, float* eta_idx, float* phi_idx, void* stream) {
    if (!coords || !regions || !workspace || !coords_pad || !eta_idx || !phi_idx) return HEPT_ERR_ARG;
    if (x && !x_pad) return HEPT_ERR_ARG;
    if (C < 2 || raw_size < 1 || N < raw_size || T < 1 || H < 1 || (x && F < 1)) return HEPT_ERR_SHAPE;
    if (workspace_bytes < hept_prepare_src_workspace_bytes(N)) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    float* keys = reinterpret_cast<float*>(ws);
    int* pos = reinterpret_cast<int*>(ws + al256((size_t)2 * N * 4));
    int* rank = reinterpret_cast<int*>(ws + 2 * al256((size_t)2 * N * 4));
    void* sort_ws = ws + 3 * al256((size_t)2 * N * 4);
    const unsigned nb = (unsigned)((N + PT - 1) / PT);
    hipLaunchKernelGGL(src_pad_keys_kernel, dim3(nb, x ? 5 : 1), dim3(PT), 0, st, x, F, coords, C, raw_size, N, keys,
                       x ? x_pad : nullptr, coords_pad);
    int rc = hept_segmented_argsort(keys, 2, N, sort_ws, pos, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(src_rank_kernel, dim3(nb, 2), dim3(PT), 0, st, pos, N, rank);
    hipLaunchKernelGGL(src_region_kernel, dim3(nb, 2 * T * H), dim3(PT), 0, st, rank, N, regions, T, H, eta_idx, phi_idx);
    return hept_launch_status();
}

// ---- the caller's only host round trip (hept_prepare_probe) -----------------------------------------------------------
// What the host has to know before it can size the outputs of hept_prepare_input: the number of clouds, the longest
// cloud, the padded length, and (for the "codes stay below 2^24" guard) the largest region count per axis -- read from
// the tensors on every call.  One workgroup: thread i finds the first point of cloud i in the sorted batch vector
// (lower bound), the cloud sizes / padded sizes are scanned in LDS, and the answers go to a host-mapped record; the
// boundaries themselves stay on the device (cloud_start, pad_start: what hept_prepare_input takes).  Replaces four
// torch launches, a device-to-host copy and a dozen host-side tensor operations per call.
constexpr int PROBE_CLOUDS = 255;   // clouds the probe resolves (boundaries 0 .. 255); more: `overflow`, caller's slow path
template <typename I>
__device__ __forceinline__ int lower_bound_of(const I* __restrict__ v, int n, long long key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)v[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void probe_kernel(const void* __restrict__ batch, int batch_i64, int n_raw, int B,
                                                    const float* __restrict__ regions, int T, int H,
                                                    int* __restrict__ cloud_start, int* __restrict__ pad_start,
                                                    int* __restrict__ host) {
    __shared__ int edge_s[257], scan_s[256];
    __shared__ float reg_s[2][256];
    const int tid = threadIdx.x;
    const int e = batch_i64 ? lower_bound_of(reinterpret_cast<const long long*>(batch), n_raw, tid)
                            : lower_bound_of(reinterpret_cast<const int*>(batch), n_raw, tid);
    edge_s[tid] = e;
    if (tid == 0) edge_s[256] = n_raw;
    // largest region count per axis (regions (T, 2, H))
    float r0 = 0.f, r1 = 0.f;
    for (int i = tid; i < T * H; i += 256) {
        const int t = i / H, h = i - t * H;
        r0 = fmaxf(r0, regions[((size_t)t * 2 + 0) * H + h]);
        r1 = fmaxf(r1, regions[((size_t)t * 2 + 1) * H + h]);
    }
    reg_s[0][tid] = r0;
    reg_s[1][tid] = r1;
    __syncthreads();
    // cloud tid exists when its first point lies inside the batch; clouds beyond the probe: overflow
    const bool exists = tid < PROBE_CLOUDS && e < n_raw;
    const int size = exists ? edge_s[tid + 1] - e : 0;      // (edge_s[255 + 1] = n_raw closes cloud 254's successor)
    const int padded = ((size + B - 1) / B) * B;
    scan_s[tid] = padded;
    __syncthreads();
    if (tid == 0) {   // 256 values: a serial scan is a few hundred cycles of one lane
        int run = 0, n_clouds = 0, longest = 0, smallest = 1 << 30;
        for (int c = 0; c < 256; ++c) {
            const int first = edge_s[c];
            const bool ex = c < PROBE_CLOUDS && first < n_raw;
            const int sz = ex ? edge_s[c + 1] - first : 0;
            pad_start[c] = run;
            cloud_start[c] = first < n_raw ? first : n_raw;
            if (ex) {
                ++n_clouds;
                longest = sz > longest ? sz : longest;
                smallest = sz < smallest ? sz : smallest;
            }
            run += scan_s[c];
        }
        pad_start[256] = run;
        cloud_start[256] = n_raw;
        float m0 = 0.f, m1 = 0.f;
        for (int i = 0; i < 256; ++i) { m0 = fmaxf(m0, reg_s[0][i]); m1 = fmaxf(m1, reg_s[1][i]); }
        host[0] = n_clouds;
        host[1] = run;                                  // n_pad
        host[2] = longest;
        host[3] = smallest;                             // < 1: a cloud id without points
        host[4] = edge_s[PROBE_CLOUDS] < n_raw ? 1 : 0; // more clouds than the probe resolves
        host[5] = __float_as_int(m0);
        host[6] = __float_as_int(m1);
        host[7] = 0x600DF00D;                           // written last: the record is complete
        __threadfence_system();
    }
}

extern "C" int hept_prepare_probe(const void* batch, int batch_is_i64, int n_raw, int B, const float* regions, int T,
                                  int H, int32_t* cloud_start, int32_t* pad_start, int32_t* host_record, void* stream) {
    if (!batch || !regions || !cloud_start || !pad_start || !host_record) return HEPT_ERR_ARG;
    if (n_raw < 1 || B < 1 || T < 1 || H < 1) return HEPT_ERR_SHAPE;
    // the record must be pinned, device-mapped host memory (hipHostMalloc / hipHostRegister): a kernel store to ordinary
    // pageable memory is a GPU fault on a machine without XNACK, not an error code -- refuse it here
    void* dev_rec = nullptr;   // the device's address of the pinned host record
    if (hipHostGetDevicePointer(&dev_rec, host_record, 0) != hipSuccess || !dev_rec) {
        (void)hipGetLastError();
        return HEPT_ERR_ARG;
    }
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, batch, batch_is_i64, n_raw, B, regions,
                       T, H, cloud_start, pad_start, reinterpret_cast<int*>(dev_rec));
    return hept_launch_status();
}

extern "C" size_t hept_prepare_workspace_bytes(int n_raw, int n_clouds, int max_cloud, int T, int H) {
    const size_t S = (size_t)2 * n_clouds;
    const size_t sortL = (size_t)max_cloud > (size_t)n_raw / 1 ? (size_t)max_cloud : (size_t)max_cloud;
    size_t b = 0;
    b += al256(S * sortL * 4);                                         // coordinate key matrix
    b += al256(S * sortL * 4);                                         // its argsort
    b += al256(S * 4);                                                 // points per (cloud, axis) segment
    b += al256((size_t)2 * n_raw * 4);                                 // ranks
    b += al256((size_t)2 * T * H * 4);                                 // row maxima
    b += al256((size_t)T * H * n_raw * 8);                             // raw codes
    b += al256((size_t)n_raw * 4) * 2;                                 // code keys + their argsort
    const size_t a1 = hept_argsort_workspace_bytes((int)S, (int)sortL);
    const size_t a2 = hept_argsort_workspace_bytes(1, n_raw);
    b += al256(a1 > a2 ? a1 : a2);
    return b;
}

// cloud_start / pad_start: device int32 arrays of n_clouds + 1 exclusive prefix sums of the raw / padded sizes.
// Outputs: pad_seq (n_pad) i64, unpad (n_pad) u8, coords_pad (n_pad, C) f32, codes_pad (T, H, n_pad) i64.
extern "C" int hept_prepare_input(const float* coords, int C, const int32_t* cloud_start, const int32_t* pad_start,
                                  int n_clouds, int n_raw, int max_cloud, int n_pad, const float* regions, int T,
                                  int H, int B, void* workspace, size_t workspace_bytes, int64_t* pad_seq,
                                  unsigned char* unpad, float* coords_pad, int64_t* codes_pad, void* stream) {
    if (!coords || !cloud_start || !pad_start || !regions || !workspace || !pad_seq || !unpad || !coords_pad ||
        !codes_pad)
        return HEPT_ERR_ARG;
    if (C < 2 || n_clouds < 1 || n_raw < 1 || max_cloud < 1 || n_pad < n_raw || T < 1 || H < 1 || B < 1)
        return HEPT_ERR_SHAPE;
    if (workspace_bytes < hept_prepare_workspace_bytes(n_raw, n_clouds, max_cloud, T, H)) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int S = 2 * n_clouds, L = max_cloud, rows = T * H;
    char* ws = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) {
        char* r = ws;
        ws += al256(bytes);
        return r;
    };
    float* keys = reinterpret_cast<float*>(take((size_t)S * L * 4));
    int* pos = reinterpret_cast<int*>(take((size_t)S * L * 4));
    int* seg_len = reinterpret_cast<int*>(take((size_t)S * 4));
    int* rank = reinterpret_cast<int*>(take((size_t)2 * n_raw * 4));
    int* row_max = reinterpret_cast<int*>(take((size_t)2 * rows * 4));
    int64_t* codes_raw = reinterpret_cast<int64_t*>(take((size_t)rows * n_raw * 8));
    float* ckeys = reinterpret_cast<float*>(take((size_t)n_raw * 4));
    int* by_code = reinterpret_cast<int*>(take((size_t)n_raw * 4));
    void* sort_ws = ws;

    const dim3 gridL((L + PT - 1) / PT, S), gridN((n_raw + PT - 1) / PT, rows);
    hipLaunchKernelGGL(coord_keys_kernel, gridL, dim3(PT), 0, st, coords, C, cloud_start, L, keys, seg_len);
    int rc = hept_segmented_argsort_ragged(keys, S, L, seg_len, sort_ws, pos, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(gridL.x, S + 1), dim3(PT), 0, st, pos, cloud_start, L, n_raw, rank, S,
                       n_clouds, regions, T, H, row_max);
    hipLaunchKernelGGL(codes_kernel, gridN, dim3(PT), 0, st, rank, cloud_start, n_clouds, n_raw, regions, T, H, row_max,
                       codes_raw, ckeys);
    rc = hept_segmented_argsort_bounded(ckeys, 1, n_raw, 0.f, 16777216.f, sort_ws, by_code, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pad_gather_kernel, dim3((n_pad + PT - 1) / PT, rows + 1), dim3(PT), 0, st, cloud_start, pad_start,
                       n_clouds, n_raw, n_pad, B, by_code, coords, C, codes_raw, rows, pad_seq, unpad, coords_pad,
                       codes_pad);
    return hept_launch_status();
}
