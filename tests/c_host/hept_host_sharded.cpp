// A PyTorch-free host for the table-sharded C ABI (include/hept_hip.h, "Table sharding over the GPUs of one node"):
// `world` processes (fork before any HIP call; here they share GPU 0, on a real node each would hipSetDevice its own
// GPU), one hept_comm each (hept_comm_create_local: the one-sided transport, no RCCL), exchange buffers allocated
// with hept_comm_p2p_alloc, their HIP IPC handles swapped over socket pairs, hept_comm_p2p_open, then
// hept_forward_sharded: every rank computes its slice of the tables and ends with the full (N, D) output.  Rank 0
// writes it.  Built and driven by tests/test_gpu_c_host.py.
//
// Problem file: as tests/c_host/hept_host.cpp (int32 header[9] = {N, H, D, C, K, T, B, precision, 0}, then the arrays).
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hept_hip.h"

#define MAX_WORLD 8

static bool xfer(int fd, void* buf, size_t n, bool send) {
    char* p = static_cast<char*>(buf);
    while (n) {
        const ssize_t k = send ? write(fd, p, n) : read(fd, p, n);
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
    }
    return true;
}

template <typename T>
static bool read_to_device(std::FILE* f, size_t count, T** dev) {
    std::vector<T> host(count);
    if (std::fread(host.data(), sizeof(T), count, f) != count) return false;
    if (hipMalloc(reinterpret_cast<void**>(dev), count * sizeof(T)) != hipSuccess) return false;
    return hipMemcpy(*dev, host.data(), count * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

// every rank sends `bytes` to rank 0, which returns the concatenation (rank order) to everybody: the "any host-side
// means" of the header -- also serves as a barrier
static bool all_gather_host(int rank, int world, int to_root, const int* from_rank, void* mine, size_t bytes, void* all) {
    if (rank == 0) {
        std::memcpy(all, mine, bytes);
        for (int r = 1; r < world; ++r)
            if (!xfer(from_rank[r], static_cast<char*>(all) + r * bytes, bytes, false)) return false;
        for (int r = 1; r < world; ++r)
            if (!xfer(from_rank[r], all, bytes * world, true)) return false;
        return true;
    }
    return xfer(to_root, mine, bytes, true) && xfer(to_root, all, bytes * world, false);
}

static int run_rank(int rank, int world, int to_root, const int* from_rank, const char* problem, const char* out_path) {
    std::FILE* f = std::fopen(problem, "rb");
    if (!f) return 1;
    int32_t hd[9];
    if (std::fread(hd, sizeof(int32_t), 9, f) != 9) return 1;
    const int N = hd[0], H = hd[1], D = hd[2], C = hd[3], K = hd[4], T = hd[5], B = hd[6], precision = hd[7];
    if (T < world) return 3;
    const int base = T / world, extra = T % world;                      // contiguous table slices, sizes differ by <= 1
    const int t0 = rank * base + (rank < extra ? rank : extra), Tl = base + (rank < extra ? 1 : 0);
    if (hipSetDevice(0) != hipSuccess) return 2;                         // one GPU per rank on a real node
    const size_t nhd = (size_t)N * H * D;
    float *q, *k, *v, *coords, *w_rpe, *alpha, *out_w, *out_b;
    int64_t* codes;
    if (!read_to_device(f, nhd, &q) || !read_to_device(f, nhd, &k) || !read_to_device(f, nhd, &v) ||
        !read_to_device(f, (size_t)N * C, &coords) || !read_to_device(f, (size_t)H * D * (C - 1) * K, &w_rpe) ||
        !read_to_device(f, (size_t)H * (D + C) * T, &alpha) || !read_to_device(f, (size_t)D * H * D, &out_w) ||
        !read_to_device(f, (size_t)D, &out_b) || !read_to_device(f, (size_t)T * H * N, &codes))
        return 2;
    std::fclose(f);
    hept_comm* comm = nullptr;
    if (hept_comm_create_local(rank, world, &comm) != HEPT_OK) return 4;
    const size_t xbytes = hept_p2p_bytes(N, H, D, world, precision);
    char mine[HEPT_IPC_HANDLE_BYTES], all[MAX_WORLD * HEPT_IPC_HANDLE_BYTES];
    if (hept_comm_p2p_alloc(comm, xbytes, mine) != HEPT_OK) {
        std::fprintf(stderr, "rank %d: p2p_alloc: %s\n", rank, hept_comm_last_error());
        return 4;
    }
    if (!all_gather_host(rank, world, to_root, from_rank, mine, HEPT_IPC_HANDLE_BYTES, all)) return 5;
    if (hept_comm_p2p_open(comm, all) != HEPT_OK) {
        std::fprintf(stderr, "rank %d: p2p_open: %s\n", rank, hept_comm_last_error());
        return 4;
    }
    char token = 1, tokens[MAX_WORLD];
    if (!all_gather_host(rank, world, to_root, from_rank, &token, 1, tokens)) return 5;   // every buffer is mapped
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) return 2;
    const size_t ws_bytes = hept_workspace_bytes(N, H, D, C, Tl, B, precision);
    const int per = (N + world - 1) / world;
    void* ws;
    float* out_full;
    if (hipMalloc(&ws, ws_bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&out_full), (size_t)per * world * D * sizeof(float)) != hipSuccess)
        return 2;
    int rc = 0;
    for (int rep = 0; rep < 3 && rc == 0; ++rep)   // epochs advance, buffers are reused
        rc = hept_forward_sharded(comm, q, k, v, coords, codes, w_rpe, alpha, out_w, out_b, N, H, D, C, K, T, t0, Tl, B,
                                  precision, /*head_groups*/ H % 4 == 0 ? 4 : 1, HEPT_TRANSPORT_ONE_SIDED, ws, ws_bytes,
                                  nullptr, 0, out_full, stream);
    if (rc != HEPT_OK) {
        std::fprintf(stderr, "rank %d: hept_forward_sharded failed: %d (%s)\n", rank, rc, hept_comm_last_error());
        return 4;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return 2;
    int status = 0;
    if (hept_comm_status(comm, &status) != HEPT_OK || status != 0) {
        std::fprintf(stderr, "rank %d: a one-sided wait timed out (status %d)\n", rank, status);
        return 6;
    }
    if (rank == 0) {
        std::vector<float> host((size_t)N * D);
        if (hipMemcpy(host.data(), out_full, host.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        std::FILE* g = std::fopen(out_path, "wb");
        if (!g || std::fwrite(host.data(), sizeof(float), host.size(), g) != host.size()) return 1;
        std::fclose(g);
    }
    if (!all_gather_host(rank, world, to_root, from_rank, &token, 1, tokens)) return 5;   // nobody unmaps early
    hept_comm_destroy(comm);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s problem.bin out.bin world\n", argv[0]);
        return 1;
    }
    const int world = std::atoi(argv[3]);
    if (world < 1 || world > MAX_WORLD) return 1;
    int from_rank[MAX_WORLD] = {0};
    pid_t kids[MAX_WORLD] = {0};
    for (int r = 1; r < world; ++r) {   // fork BEFORE the first HIP call of this process
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 1;
        const pid_t pid = fork();
        if (pid < 0) return 1;
        if (pid == 0) {
            close(sv[0]);
            _exit(run_rank(r, world, sv[1], nullptr, argv[1], argv[2]));
        }
        close(sv[1]);
        from_rank[r] = sv[0];
        kids[r] = pid;
    }
    int rc = run_rank(0, world, -1, from_rank, argv[1], argv[2]);
    for (int r = 1; r < world; ++r) {
        int st = 0;
        waitpid(kids[r], &st, 0);
        if (rc == 0 && !(WIFEXITED(st) && WEXITSTATUS(st) == 0)) rc = 10 + r;
    }
    return rc;
}
