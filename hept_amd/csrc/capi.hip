// Whole-operator entry points of the C ABI (include/hept_hip.h): workspace carving + launch order.
// Replaces HEPTAttention.forward, reference example/hept.py:43-81.
#include "common.h"
#include "comm.h"
#include "p2p_dev.h"

namespace {

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
    void* qhat;
    void* kvhat;
    float* qproj;
    float* kproj;
    float* minmax;
    int32_t* pos;   // (2, Tl, H, N): q then k
    int32_t* pos_chunk;  // Tl > HEPT_MAX_TABLES only: (2, chunk, H, N) the sort of one chunk writes before it is filed
    void* sort_ws;
    float* part;    // (Tl, N, H, 32)
    size_t bytes;
};

// proj / minmax / sort scratch hold one chunk of at most HEPT_MAX_TABLES tables (prep_hash and the sort work on such
// chunks); the permutations and the partial rows exist for all Tl tables of the call
inline int chunk_tables(int Tl) { return Tl < HEPT_MAX_TABLES ? Tl : HEPT_MAX_TABLES; }

Workspace carve(void* base, int N, int H, int C, int Tl, int precision) {
    const size_t esz = (precision == HEPT_PREC_F32 || precision == HEPT_PREC_F32_MFMA || precision == HEPT_PREC_F32_DIFF) ? 4 : 2;
    const int Tc = chunk_tables(Tl);
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* r = p ? p + off : nullptr;
        off += up256(bytes);
        return r;
    };
    Workspace w;
    w.qhat = take((size_t)H * N * 32 * esz);
    w.kvhat = take((size_t)H * N * 64 * esz);
    w.qproj = reinterpret_cast<float*>(take((size_t)Tc * H * N * 4));
    w.kproj = reinterpret_cast<float*>(take((size_t)Tc * H * N * 4));
    w.minmax = reinterpret_cast<float*>(take((size_t)HEPT_PREP_GRID * Tc * H * 4 * 4));
    w.pos = reinterpret_cast<int32_t*>(take((size_t)2 * Tl * H * N * 4));
    w.pos_chunk = Tl > Tc ? reinterpret_cast<int32_t*>(take((size_t)2 * Tc * H * N * 4)) : nullptr;
    w.sort_ws = take(hept_sort_workspace_bytes(N, H, Tc));
    w.part = reinterpret_cast<float*>(take((size_t)Tl * N * H * 32 * 4));
    w.bytes = off;
    return w;
}

// ---- optional stage timing (hept_profile_*): a pool of HIP events recorded on the caller's stream
constexpr int PROF_SLOTS = 8;   // events per call: 6 stages (the table-sharded call has two more than hept_forward) + the
                                // mark between the sort's two launches (PROF_SORT_MID, outside the chain of stages)
constexpr int PROF_SORT_MID = 7;
struct Profiler {
    int mode = 0, max_calls = 0, n_calls = 0;
    int stride = 1, seen = 0;  // only every `stride`-th forward call is bracketed
    hipEvent_t* ev = nullptr;  // [max_calls][PROF_SLOTS]
    int* last = nullptr;       // [max_calls] highest slot recorded in the call
    int* mid = nullptr;        // [max_calls] the call recorded PROF_SORT_MID
} g_prof;

inline bool prof_active() {
    return g_prof.mode != 0 && g_prof.n_calls < g_prof.max_calls && g_prof.seen % g_prof.stride == 0;
}
inline void prof_mark(int slot, hipStream_t st) {
    if (!prof_active()) return;
    if (g_prof.mode == 1 && slot != 2 && slot != 3) return;
    (void)hipEventRecord(g_prof.ev[(size_t)g_prof.n_calls * PROF_SLOTS + slot], st);
    if (slot == PROF_SORT_MID) {
        g_prof.mid[g_prof.n_calls] = 1;
        return;
    }
    int& last = g_prof.last[g_prof.n_calls];
    const bool opens = slot == (g_prof.mode == 1 ? 2 : 0);   // first event of a call: the pool entry is reused
    if (opens) g_prof.mid[g_prof.n_calls] = 0;
    if (opens || slot > last) last = slot;
}
inline void prof_call_done() {
    if (prof_active()) ++g_prof.n_calls;
    if (g_prof.mode != 0) ++g_prof.seen;
}

// region shift of the reference's src variant (src/models/attention/hept.py:46-56, 89-101); eta == nullptr: the
// example variant with its integer AND codes
struct GeoShift {
    const float* eta = nullptr;
    const float* phi = nullptr;
    const float* cfac = nullptr;
    int raw_size = -1;
};

// HEPT_NO_ROW_RIDERS=1: the row builder keeps its v role (A/B measurements; read once)
inline bool row_riders_off() {
    static const bool off = [] { const char* e = getenv("HEPT_NO_ROW_RIDERS"); return e && *e && *e != '0'; }();
    return off;
}

// Value rows read from the caller's v (run_begin: `dv`) pay where the block attention is bound by issue, not by its
// gathers: blocks above 128 points (pileup batch, f32 rows: 413.9 -> 405.7 us per forward; at B = 128 the second gather
// stream costs the kernel what the riders cost the sort, 294 vs 294-298 us, and short clouds lose 1-3 us --
// profiles/r05_experiments.txt)
static inline bool direct_v_pays(int B) { return B > 128; }

// everything before the block attention, for tables [t0, t0 + Tl): parameter math, augmented rows + hashes, sort.
// Leaves qhat / kvhat and the permutations (w.pos: q then k, (Tl, H, N) each) in the workspace.
int run_begin(const float* q, const float* k, const float* v, const float* coords, const int64_t* codes,
              const GeoShift& geo, const float* w_rpe, const float* alpha, int N, int H, int D, int C, int K, int T,
              int t0, int Tl, int precision, const Workspace& w, void* stream, VSrc* dv = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    prof_mark(0, st);
    // K > 0: `w_rpe` is w_rpe.weight and the row builder computes sqrt_w (H, C) from it in its prologue, every call
    // (reference example/hept.py:22-25; nothing is cached, so an in-place update of the parameter is always seen);
    // K == 0: the caller passes sqrt_w itself (hept_rpe_scale)
    int rc = HEPT_OK;
    int32_t* qpos = w.pos;
    int32_t* kpos = w.pos + (size_t)Tl * H * N;
    // The v half of the kvhat rows does not depend on anything this call computes: when the sort has a bucket-sort launch
    // (segments longer than the one-workgroup sort) that launch -- a latency-bound chain that leaves the memory system
    // idle -- carries it as rider workgroups, and the row builder runs its q and k roles only (sort_tables.hip: RowsJob)
    const int raw_size = geo.eta ? geo.raw_size : N;
    // ... unless the launch is too short to hide them: with ONE local table (BASELINE config 4: one table per GPU) and
    // 16-bit rows the bucket sort is ~12 us of its own work against ~20 us of riders -- the row builder keeps its v
    // role there (tracking-60k, T = 1: 111.0 -> 104.7 us per forward; two tables and more, and f32 rows at any count,
    // are faster with riders: profiles/r04_experiments.txt)
    const bool f32_rows = precision == HEPT_PREC_F32 || precision == HEPT_PREC_F32_MFMA || precision == HEPT_PREC_F32_DIFF;
    // (HEPT_FORCE_ROW_RIDERS=1: riders at any table count -- A/B measurements; read once)
    static const bool force_ride = [] { const char* e = getenv("HEPT_FORCE_ROW_RIDERS"); return e && *e && *e != '0'; }();
    // f32 rows, round 5: nobody builds the v half at all when the caller of run_begin runs the block attention itself
    // (`dv`): the split-bf16 kernel stages its value planes from the caller's v rows (96 of 128 fetched bytes used, the
    // same sectors as a padded kvhat row) and the bucket sort loses its riders -- 46 MB read + 61.5 MB written per call
    // at tracking-60k (HEPT_NO_DIRECT_V=1: A/B measurements; read once)
    static const bool no_direct_v = [] { const char* e = getenv("HEPT_NO_DIRECT_V"); return e && *e && *e != '0'; }();
    const bool direct_v = dv && precision == HEPT_PREC_F32 && D % 4 == 0 && !no_direct_v &&
                          (reinterpret_cast<uintptr_t>(v) & 15) == 0;
    if (dv) *dv = direct_v ? VSrc{v, raw_size} : VSrc{};
    const bool ride = !direct_v && Tl <= HEPT_MAX_TABLES && (Tl >= 2 || f32_rows || force_ride) && hept_sort_carries_rows(N, H, D) && !row_riders_off();
    const HeptRowsJob job{v, w.kvhat, N, raw_size, H, D, precision};
    const HeptRowsJob* rows = ride ? &job : nullptr;
    for (int c0 = 0; c0 < Tl; c0 += HEPT_MAX_TABLES) {   // chunks of tables (the rows are rewritten identically)
        const int tc = Tl - c0 < HEPT_MAX_TABLES ? Tl - c0 : HEPT_MAX_TABLES;
        // (the row builder also clears the sort's bucket counters on its way: no fill launch in front of the sort)
        void* zptr = nullptr;
        size_t zbytes = 0;
        hept_sort_zero_block(w.sort_ws, N, H, tc, &zptr, &zbytes);
        rc = hept_prep_hash_rpe(q, k, v, coords, w_rpe, K, alpha, codes, N, raw_size, H, D, C, T,
                                t0 + c0, tc, precision, w.qhat, w.kvhat, w.qproj, w.kproj, w.minmax, stream,
                                (ride || direct_v) ? 2 : 3, zptr, zbytes);
        if (rc) return rc;
        if (c0 == 0) prof_mark(1, st);
        // the sort writes one (2, tc, H, N) array: straight into w.pos when the call is a single chunk
        int32_t* cq = Tl <= HEPT_MAX_TABLES ? qpos : w.pos_chunk;
        int32_t* ck = cq + (size_t)tc * H * N;
        rc = geo.eta ? hept_sort_tables_src_rows(w.qproj, w.kproj, geo.eta, geo.phi, geo.cfac, w.minmax, N, H, T, t0 + c0,
                                                 tc, w.sort_ws, cq, ck, rows, stream, zbytes != 0)
                     : hept_sort_tables_rows(w.qproj, w.kproj, codes, w.minmax, N, H, T, t0 + c0, tc, w.sort_ws, cq, ck,
                                             rows, stream, zbytes != 0);
        if (rc) return rc;
        if (cq != qpos) {
            const size_t off = (size_t)c0 * H * N, bytes = (size_t)tc * H * N * 4;
            if (hipMemcpyAsync(qpos + off, cq, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(kpos + off, ck, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return HEPT_ERR_LAUNCH;
        }
    }
    prof_mark(2, st);
    return HEPT_OK;
}

// stages shared by hept_forward / hept_forward_partial; leaves per-table partials in `part`
int run_tables(const float* q, const float* k, const float* v, const float* coords, const int64_t* codes,
               const GeoShift& geo, const float* w_rpe, const float* alpha, int N, int H, int D, int C, int K, int T,
               int t0, int Tl, int B, int precision, const Workspace& w, float* part, void* stream) {
    VSrc dv;
    int rc = run_begin(q, k, v, coords, codes, geo, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, precision, w, stream,
                       direct_v_pays(B) ? &dv : nullptr);
    if (rc) return rc;
    rc = hept_block_attn_heads_push(w.qhat, w.kvhat, w.pos, w.pos + (size_t)Tl * H * N, N, H, D, Tl, B, precision, 0, H, H,
                                    0, N, part, nullptr, stream, dv);
    prof_mark(3, (hipStream_t)stream);
    return rc;
}

}  // namespace

// internal (common.h): the sort marks the boundary between its two launches (mode 2 only; first chunk of tables only)
void hept_prof_mark_sort_mid(void* stream) {
    if (g_prof.mode == 2 && prof_active() && !g_prof.mid[g_prof.n_calls]) prof_mark(PROF_SORT_MID, (hipStream_t)stream);
}

extern "C" int hept_abi_version(void) { return 20; }

extern "C" int hept_part_precision(int precision, int D) {
    return (precision != HEPT_PREC_F32 && precision != HEPT_PREC_F32_MFMA && precision != HEPT_PREC_F32_DIFF && D == 24) ? HEPT_PREC_BF16 : HEPT_PREC_F32;
}

extern "C" int hept_check_shape(int N, int H, int D, int C, int Tl, int B) {
    if (N < 1 || B < 1 || B > HEPT_MAX_BLOCK || N % B != 0) return HEPT_ERR_SHAPE;
    if (Tl < 1) return HEPT_ERR_SHAPE;  // any number of tables: prep_hash and the sort run in chunks of HEPT_MAX_TABLES
    // rows are 32 columns wide: [D features | C scaled coordinates | 0.. | norm], denominators ride at column D.
    // (H = 8 with the shipped models' (D, C) pairs takes the tuned row builder, anything else the generic one.)
    if (H < 1 || H > 16 || D < 1 || D > 27 || C < 2 || D + C > 30) return HEPT_ERR_SHAPE;  // (D <= 27: the combine reads the 28 leading floats of a row, denominator at column D)
    return HEPT_OK;
}

extern "C" size_t hept_workspace_bytes(int N, int H, int D, int C, int Tl, int B, int precision) {
    (void)D;
    (void)B;
    return carve(nullptr, N, H, C, Tl, precision).bytes;
}

namespace {
int forward_impl(const float* q, const float* k, const float* v, const float* coords, const int64_t* codes,
                 const GeoShift& geo, const float* w_rpe, const float* alpha, const float* out_weight,
                 const float* out_bias, int N, int H, int D, int C, int K, int T, int B, int precision,
                 void* workspace, size_t workspace_bytes, float* out, void* stream) {
    if (!q || !k || !v || !coords || !w_rpe || !alpha || !out_weight || !workspace || !out) return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, T, B);
    if (rc) return rc;
    const Workspace w = carve(workspace, N, H, C, T, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    rc = run_tables(q, k, v, coords, codes, geo, w_rpe, alpha, N, H, D, C, K, T, 0, T, B, precision, w, w.part,
                    stream);
    if (rc) return rc;
    rc = hept_combine_out(w.part, hept_part_precision(precision, D), T, N, H, D, 0, N, out_weight, out_bias, out,
                          stream);
    prof_mark(4, (hipStream_t)stream);
    prof_call_done();
    return rc;
}

int forward_partial_impl(const float* q, const float* k, const float* v, const float* coords, const int64_t* codes,
                         const GeoShift& geo, const float* w_rpe, const float* alpha, int N, int H, int D, int C,
                         int K, int T, int t0, int Tl, int B, int precision, int acc_precision, void* workspace,
                         size_t workspace_bytes, float* acc, void* stream) {
    if (!q || !k || !v || !coords || !w_rpe || !alpha || !workspace || !acc) return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, Tl, B);
    if (rc) return rc;
    if (t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (acc_precision != HEPT_PREC_F32 && acc_precision != hept_part_precision(precision, D)) return HEPT_ERR_SHAPE;
    const Workspace w = carve(workspace, N, H, C, Tl, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    // one local table already in the requested row format: block_attn scatters straight into acc, no reduction pass
    const int pprec = hept_part_precision(precision, D);
    const bool direct = Tl == 1 && pprec == acc_precision;
    float* part = direct ? acc : w.part;
    rc = run_tables(q, k, v, coords, codes, geo, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, B, precision, w, part,
                    stream);
    if (!rc && !direct) rc = hept_reduce_tables(w.part, pprec, Tl, N, H, D, acc, acc_precision, stream);
    prof_mark(4, (hipStream_t)stream);
    prof_call_done();
    return rc;
}
}  // namespace

extern "C" int hept_forward(const float* q, const float* k, const float* v, const float* coords,
                            const int64_t* codes, const float* w_rpe, const float* alpha, const float* out_weight,
                            const float* out_bias, int N, int H, int D, int C, int K, int T, int B, int precision,
                            void* workspace, size_t workspace_bytes, float* out, void* stream) {
    if (!codes) return HEPT_ERR_ARG;
    return forward_impl(q, k, v, coords, codes, GeoShift{}, w_rpe, alpha, out_weight, out_bias, N, H, D, C, K, T, B,
                        precision, workspace, workspace_bytes, out, stream);
}

extern "C" int hept_forward_partial(const float* q, const float* k, const float* v, const float* coords,
                                    const int64_t* codes, const float* w_rpe, const float* alpha, int N, int H,
                                    int D, int C, int K, int T, int t0, int Tl, int B, int precision,
                                    int acc_precision, void* workspace, size_t workspace_bytes, float* acc,
                                    void* stream) {
    if (!codes) return HEPT_ERR_ARG;
    return forward_partial_impl(q, k, v, coords, codes, GeoShift{}, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, B,
                                precision, acc_precision, workspace, workspace_bytes, acc, stream);
}

extern "C" int hept_forward_src(const float* q, const float* k, const float* v, const float* coords,
                                const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                                const float* w_rpe, const float* alpha, const float* out_weight,
                                const float* out_bias, int N, int H, int D, int C, int K, int T, int B,
                                int precision, void* workspace, size_t workspace_bytes, float* out, void* stream) {
    if (!eta_idx || !phi_idx || !cfac) return HEPT_ERR_ARG;
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    return forward_impl(q, k, v, coords, nullptr, GeoShift{eta_idx, phi_idx, cfac, raw_size}, w_rpe, alpha,
                        out_weight, out_bias, N, H, D, C, K, T, B, precision, workspace, workspace_bytes, out, stream);
}

extern "C" int hept_forward_partial_src(const float* q, const float* k, const float* v, const float* coords,
                                        const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                                        const float* w_rpe, const float* alpha, int N, int H, int D, int C, int K,
                                        int T, int t0, int Tl, int B, int precision, int acc_precision,
                                        void* workspace, size_t workspace_bytes, float* acc, void* stream) {
    if (!eta_idx || !phi_idx || !cfac) return HEPT_ERR_ARG;
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    return forward_partial_impl(q, k, v, coords, nullptr, GeoShift{eta_idx, phi_idx, cfac, raw_size}, w_rpe, alpha, N,
                                H, D, C, K, T, t0, Tl, B, precision, acc_precision, workspace, workspace_bytes, acc,
                                stream);
}

namespace {
int partial_begin_impl(const float* q, const float* k, const float* v, const float* coords, const int64_t* codes,
                       const GeoShift& geo, const float* w_rpe, const float* alpha, int N, int H, int D, int C, int K,
                       int T, int t0, int Tl, int B, int precision, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (!q || !k || !v || !coords || !w_rpe || !alpha || !workspace) return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, Tl, B);
    if (rc) return rc;
    if (t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    const Workspace w = carve(workspace, N, H, C, Tl, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    return run_begin(q, k, v, coords, codes, geo, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, precision, w, stream);
}
}  // namespace

extern "C" int hept_partial_begin(const float* q, const float* k, const float* v, const float* coords,
                                  const int64_t* codes, const float* w_rpe, const float* alpha, int N, int H, int D,
                                  int C, int K, int T, int t0, int Tl, int B, int precision, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!codes) return HEPT_ERR_ARG;
    return partial_begin_impl(q, k, v, coords, codes, GeoShift{}, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, B, precision,
                              workspace, workspace_bytes, stream);
}

extern "C" int hept_partial_begin_src(const float* q, const float* k, const float* v, const float* coords,
                                      const float* eta_idx, const float* phi_idx, const float* cfac, int raw_size,
                                      const float* w_rpe, const float* alpha, int N, int H, int D, int C, int K, int T,
                                      int t0, int Tl, int B, int precision, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (!eta_idx || !phi_idx || !cfac) return HEPT_ERR_ARG;
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    return partial_begin_impl(q, k, v, coords, nullptr, GeoShift{eta_idx, phi_idx, cfac, raw_size}, w_rpe, alpha, N, H,
                              D, C, K, T, t0, Tl, B, precision, workspace, workspace_bytes, stream);
}

namespace {
// block attention of heads [h0, h0 + hg) for the tables of the preceding run_begin, summed over those tables into
// dst (n_pad, hg, row); rows [N, n_pad) are zero
int partial_heads_impl(const Workspace& w, int N, int H, int D, int Tl, int B, int precision, int h0, int hg, int n_pad,
                       int acc_precision, float* dst, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int pprec = hept_part_precision(precision, D);
    const int32_t* qpos = w.pos;
    const int32_t* kpos = w.pos + (size_t)Tl * H * N;
    // one local table already in the requested row format: block_attn scatters straight into the group's rows of
    // dst; the padding rows are zeroed (nothing reads them, but they travel)
    const bool direct = Tl == 1 && pprec == acc_precision;
    const bool rec = g_prof.mode == 1;
    int rc;
    if (rec) prof_mark(2, st);
    if (direct) {
        const size_t row_bytes = (size_t)hg * (acc_precision == HEPT_PREC_BF16 ? 64 : 128);
        if (n_pad > N && hipMemsetAsync(reinterpret_cast<char*>(dst) + (size_t)N * row_bytes, 0,
                                        (size_t)(n_pad - N) * row_bytes, st) != hipSuccess)
            return HEPT_ERR_LAUNCH;
        rc = hept_block_attn_heads(w.qhat, w.kvhat, qpos, kpos, N, H, D, Tl, B, precision, h0, hg, hg, h0, n_pad, dst,
                                   stream);
        if (rec) prof_mark(3, st);
    } else {
        rc = hept_block_attn_heads(w.qhat, w.kvhat, qpos, kpos, N, H, D, Tl, B, precision, h0, hg, H, 0, N, w.part,
                                   stream);
        if (rec) prof_mark(3, st);
        if (!rc) rc = hept_reduce_heads(w.part, pprec, Tl, N, H, D, h0, hg, n_pad, dst, acc_precision, stream);
    }
    if (rec) prof_call_done();
    return rc;
}
}  // namespace

extern "C" int hept_partial_heads(void* workspace, size_t workspace_bytes, int N, int H, int D, int C, int Tl, int B,
                                  int precision, int h0, int hg, int n_pad, int acc_precision, float* dst,
                                  void* stream) {
    if (!workspace || !dst) return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, Tl, B);
    if (rc) return rc;
    if (h0 < 0 || hg < 1 || h0 + hg > H || n_pad < N) return HEPT_ERR_SHAPE;
    if (acc_precision != HEPT_PREC_F32 && acc_precision != hept_part_precision(precision, D)) return HEPT_ERR_SHAPE;
    const Workspace w = carve(workspace, N, H, C, Tl, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    return partial_heads_impl(w, N, H, D, Tl, B, precision, h0, hg, n_pad, acc_precision, dst, stream);
}

// ---- table sharding in one call: kernels on the caller's stream, RCCL transfers of finished head groups on the
//      communicator's side stream
extern "C" size_t hept_exchange_bytes(int N, int H, int D, int world, int precision) {
    if (N < 1 || H < 1 || world < 1) return 0;
    const size_t per = ((size_t)N + world - 1) / world;
    const size_t row = hept_part_precision(precision, D) == HEPT_PREC_BF16 ? 64 : 128;
    return 2 * up256(per * world * H * row);  // send + recv, all head groups
}

namespace {
int sharded_steps(hept_comm* comm, const float* q, const float* k, const float* v, const float* coords,
                  const int64_t* codes, const GeoShift& geo, const float* w_rpe, const float* alpha,
                  const float* out_weight, const float* out_bias, int N, int H, int D, int C, int K, int T, int t0,
                  int Tl, int B, int precision, int head_groups, bool one_sided, const Workspace& w, void* xbuf,
                  float* out_full, void* stream) {
    P2pLayout lay = hept_p2p_layout(N, H, D, comm->world, precision);
    // view mode: the gathered output of step e lies in region e & 1 and is read there by the caller
    const bool view = one_sided && comm->out_view;
    if (view && (comm->epoch & 1u)) lay.out_off += lay.out_bytes;
    hipStream_t st = (hipStream_t)stream;
    const int world = comm->world, hg = H / head_groups;
    const int per = (N + world - 1) / world, n_pad = per * world;
    const int aprec = hept_part_precision(precision, D);
    const size_t row = aprec == HEPT_PREC_BF16 ? 64 : 128;
    const size_t group_bytes = (size_t)n_pad * hg * row;       // one head group, all ranks' slices
    // this rank's points [rank * per, ...): the `world` received slices are the "tables" of the combine
    const int first = comm->rank * per;
    const int cnt = first >= N ? 0 : (N - first < per ? N - first : per);
    // one-sided transport with the fused combine: the rows this rank owes ITSELF stay in ordinary device memory
    // (hept_comm::p2p_self) instead of the uncached buffer -- 1 / world of the rows at cached speed on both sides
    const bool mirror = one_sided && D == 24 && cnt >= 1;
    char* send = reinterpret_cast<char*>(xbuf);
    char* recv = one_sided ? comm->p2p_local + lay.recv_off : send + up256((size_t)n_pad * H * row);
    VSrc dv;
    int rc = run_begin(q, k, v, coords, codes, geo, w_rpe, alpha, N, H, D, C, K, T, t0, Tl, precision, w, stream,
                       direct_v_pays(B) ? &dv : nullptr);
    if (rc) return rc;
    const int pprec = hept_part_precision(precision, D);
    const bool direct = Tl == 1 && !one_sided;   // one local table: block_attn scatters straight into the send buffer
    const int32_t* qpos = w.pos;
    const int32_t* kpos = w.pos + (size_t)Tl * H * N;
    const bool rec = g_prof.mode == 1;
    const bool rec_all = g_prof.mode == 2;   // stage times of the whole sharded step (bench.py: exchange terms)
    if (one_sided && Tl == 1) {
        // ONE local table (BASELINE config 4): nothing to sum, so nothing to carry -- the block attention of a head
        // group stores every finished row straight into the receive buffer of the rank that owns the point (16-B
        // pieces of 64-B rows, system-scope stores) and its last workgroup raises the flags.  The rows cross the
        // links while the kernel runs; only the drain of the last stores is exposed.
        for (int g = 0; g < head_groups; ++g) {
            PushArgs pa;
            rc = hept_p2p_direct_args(comm, N, H, D, g * hg, hg, g, aprec, lay, mirror, &pa);
            if (rc) return rc;
            if (rec) prof_mark(2, st);
            rc = hept_block_attn_heads_push(w.qhat, w.kvhat, qpos, kpos, N, H, D, 1, B, precision, g * hg, hg, hg, g * hg,
                                            n_pad, nullptr, &pa, stream, dv);
            if (rec) {
                prof_mark(3, st);
                prof_call_done();
            }
            if (rc) return rc;
        }
        if (rec_all) {
            prof_mark(3, st);   // end of the (last) attention launch = the rows have been stored and flagged
            prof_mark(4, st);   // no separate push
        }
    } else if (one_sided) {
        // ONE stream, no events: the launch that computes head group g also carries, as its first workgroups, the
        // table sum + one-sided push of group g - 1 (link-bound work beside gather-bound work); the last group's
        // push has nothing left to ride on and runs alone.
        constexpr int PUSH_WGS = 256;   // one per CU: enough to keep the seven links busy (multiple of 8: XCD map)
        for (int g = 0; g < head_groups; ++g) {
            PushArgs pa;
            const bool carry = g > 0;
            if (carry) {
                rc = hept_p2p_push_args(comm, w.part, pprec, Tl, N, H, D, (g - 1) * hg, hg, g - 1, aprec, lay, PUSH_WGS,
                                        mirror, &pa);
                if (rc) return rc;
            }
            if (rec) prof_mark(2, st);
            rc = hept_block_attn_heads_push(w.qhat, w.kvhat, qpos, kpos, N, H, D, Tl, B, precision, g * hg, hg, H, 0, N,
                                            w.part, carry ? &pa : nullptr, stream, dv);
            if (rec) {
                prof_mark(3, st);
                prof_call_done();
            }
            if (rc) return rc;
        }
        if (rec_all) prof_mark(3, st);
        rc = hept_p2p_reduce_push(comm, w.part, pprec, Tl, N, H, D, (head_groups - 1) * hg, hg, head_groups - 1, aprec,
                                  lay, mirror, st);
        if (rc) return rc;
        if (rec_all) prof_mark(4, st);   // the exposed push: table sum + rows of the last head group
    }
    // RCCL transport.  Caller's stream: the block attention of the head groups back to back.  Side stream: for every
    // group but the last, wait for its attention, sum the local tables into the send buffer and put it on the links.
    // The last group has nothing left to hide behind: its table sum and transfer run on the caller's stream (a
    // cross-stream hand-over costs ~15 us each way), which then joins the side stream and combines.
    for (int g = 0; !one_sided && g < head_groups; ++g) {
        const bool last = g == head_groups - 1;
        float* dst = reinterpret_cast<float*>(send + g * group_bytes);
        if (rec) prof_mark(2, st);
        if (direct) {
            const size_t row_bytes = (size_t)hg * row;
            if (n_pad > N && hipMemsetAsync(reinterpret_cast<char*>(dst) + (size_t)N * row_bytes, 0,
                                            (size_t)(n_pad - N) * row_bytes, st) != hipSuccess)
                return HEPT_ERR_LAUNCH;
            rc = hept_block_attn_heads_push(w.qhat, w.kvhat, qpos, kpos, N, H, D, Tl, B, precision, g * hg, hg, hg, g * hg,
                                            n_pad, dst, nullptr, stream, dv);
        } else {
            rc = hept_block_attn_heads_push(w.qhat, w.kvhat, qpos, kpos, N, H, D, Tl, B, precision, g * hg, hg, H, 0, N,
                                            w.part, nullptr, stream, dv);
        }
        if (rec) {
            prof_mark(3, st);
            prof_call_done();
        }
        if (rc) return rc;
        if (rec_all && last) prof_mark(3, st);
        hipStream_t xs = last ? st : comm->side;
        if (!last && (hipEventRecord(comm->fork[g], st) != hipSuccess ||
                      hipStreamWaitEvent(comm->side, comm->fork[g], 0) != hipSuccess))
            return HEPT_ERR_LAUNCH;
        if (!direct) {
            rc = hept_reduce_heads(w.part, pprec, Tl, N, H, D, g * hg, hg, n_pad, dst, aprec, xs);
            if (rc) return rc;
        }
        rc = hept_comm_all_to_all(comm, dst, recv + g * group_bytes, (size_t)per * hg * row, xs);
        if (rc) return rc;
    }
    if (!one_sided && head_groups > 1 &&
        (hipEventRecord(comm->join, comm->side) != hipSuccess || hipStreamWaitEvent(st, comm->join, 0) != hipSuccess))
        return HEPT_ERR_LAUNCH;
    if (rec_all && !one_sided) prof_mark(4, st);   // the exposed transfer: last group's table sum + all-to-all (+ join)
    if (mirror) {
        // wait for the rows + combine + push of the output slice + output flag in one kernel, then gather
        // (view mode: the gathered output in the exchange buffer must be whole, so this rank's own slice goes there
        //  too -- and from there into out_full, if the caller passed one all the same)
        rc = hept_p2p_combine_push(comm, head_groups, per, cnt, H, hg, aprec, out_weight, out_bias, lay,
                                   view ? nullptr : out_full, st);
        if (rc) return rc;
        if (rec_all) prof_mark(5, st);   // wait for the rows + combine + output slice to every rank
        rc = out_full ? hept_p2p_wait_copy_out(comm, n_pad, N, D, lay, out_full, !view, st) : hept_p2p_wait_out(comm, N, D, lay, st);
        if (!rc && view) comm->last_out = reinterpret_cast<const float*>(comm->p2p_local + lay.out_off);
        if (rec_all) {
            prof_mark(6, st);            // wait for every rank's slice + copy out
            prof_call_done();
        }
        return rc;
    }
    if (one_sided) {
        rc = hept_p2p_wait_rows(comm, head_groups, st);
        if (rc) return rc;
    }
    // one-sided: the slice is produced in this rank's own output region, pushed to the others, and the gathered
    // output is copied out once every rank's slice has arrived
    float* gathered = one_sided ? reinterpret_cast<float*>(comm->p2p_local + lay.out_off) : out_full;
    float* mine = gathered + (size_t)first * D;
    rc = hept_combine_groups(reinterpret_cast<const float*>(recv), aprec, world, per, H, D, 0, cnt, hg,
                             group_bytes / 4, out_weight, out_bias, mine, stream);
    if (rc) return rc;
    if (cnt < per && hipMemsetAsync(mine + (size_t)cnt * D, 0, (size_t)(per - cnt) * D * 4, st) != hipSuccess)
        return HEPT_ERR_LAUNCH;
    if (one_sided) {
        rc = hept_p2p_push_out(comm, per, D, lay, st);
        if (rc) return rc;
        if (rec_all) prof_mark(5, st);
        rc = out_full ? hept_p2p_wait_copy_out(comm, n_pad, N, D, lay, out_full, false, st) : hept_p2p_wait_out(comm, N, D, lay, st);
        if (!rc && view) comm->last_out = reinterpret_cast<const float*>(comm->p2p_local + lay.out_off);
    } else {
        if (rec_all) prof_mark(5, st);
        rc = hept_comm_all_gather_f32(comm, out_full, (size_t)per * D, st);
    }
    if (rec_all) {
        prof_mark(6, st);
        prof_call_done();
    }
    return rc;
}

int forward_sharded_impl(hept_comm* comm, const float* q, const float* k, const float* v, const float* coords,
                         const int64_t* codes, const GeoShift& geo, const float* w_rpe, const float* alpha,
                         const float* out_weight, const float* out_bias, int N, int H, int D, int C, int K, int T,
                         int t0, int Tl, int B, int precision, int head_groups, int transport, void* workspace,
                         size_t workspace_bytes, void* xbuf, size_t xbuf_bytes, float* out_full, void* stream) {
    // Everything that can be refused is refused BEFORE the step takes its epoch: a rank that bails out here has not
    // moved, and the others time out against it once (and say so) instead of running one epoch apart for good.
    if (!comm || !q || !k || !v || !coords || !w_rpe || !alpha || !out_weight || !workspace) return HEPT_ERR_ARG;
    const bool one_sided = transport == HEPT_TRANSPORT_ONE_SIDED;
    // no output pointer: the one-sided transport in view mode only (the output is read in the exchange buffer)
    if (!out_full && !(one_sided && comm->out_view)) return HEPT_ERR_ARG;
    if (!one_sided && (transport != HEPT_TRANSPORT_RCCL || !xbuf)) return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, Tl, B);
    if (rc) return rc;
    if (t0 < 0 || t0 + Tl > T) return HEPT_ERR_SHAPE;
    if (head_groups < 1 || head_groups > HEPT_MAX_HEAD_GROUPS || H % head_groups != 0) return HEPT_ERR_SHAPE;
    if (K < 0 || (K > 0 && H * (C - 1) * K > 1024)) return HEPT_ERR_SHAPE;
    const Workspace w = carve(workspace, N, H, C, Tl, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    if (!one_sided && xbuf_bytes < hept_exchange_bytes(N, H, D, comm->world, precision)) return HEPT_ERR_ARG;
    if (one_sided) {
        const P2pLayout lay = hept_p2p_layout(N, H, D, comm->world, precision);
        if (!comm->p2p_open || comm->p2p_bytes < lay.bytes) return HEPT_ERR_ARG;
        if (head_groups * comm->world > 256) return HEPT_ERR_SHAPE;
        // a wait of an EARLIER step timed out on this GPU (the kernel wrote the host-mapped status word; that step's
        // output was NaN), or an earlier step failed after taking its epoch: the transport is out of step with its
        // peers and refuses to run until every rank has called hept_comm_reset_status (TableSharding.check does)
        if (const int bad = hept_p2p_failed(comm)) {
            hept_comm_set_error("one-sided exchange",
                                bad & 4 ? "an earlier step failed after taking its epoch; the transport needs hept_comm_reset_status on every rank"
                                        : "a wait for a peer timed out in an earlier step (its output was NaN); the transport needs hept_comm_reset_status on every rank");
            return HEPT_ERR_COMM;
        }
        ++comm->epoch;
    }
    rc = sharded_steps(comm, q, k, v, coords, codes, geo, w_rpe, alpha, out_weight, out_bias, N, H, D, C, K, T, t0, Tl, B,
                       precision, head_groups, one_sided, w, xbuf, out_full, stream);
    // a failure after the epoch was taken (a launch error half-way through the step): the peers will time out on this
    // step, and this rank may have raised only some of its flags -- fatal for the transport until it is reset
    if (rc && one_sided) comm->broken = true;
    return rc;
}
}  // namespace

extern "C" int hept_forward_sharded(hept_comm* comm, const float* q, const float* k, const float* v,
                                    const float* coords, const int64_t* codes, const float* w_rpe,
                                    const float* alpha, const float* out_weight, const float* out_bias, int N, int H,
                                    int D, int C, int K, int T, int t0, int Tl, int B, int precision, int head_groups,
                                    int transport, void* workspace, size_t workspace_bytes, void* xbuf,
                                    size_t xbuf_bytes, float* out_full, void* stream) {
    if (!codes) return HEPT_ERR_ARG;
    return forward_sharded_impl(comm, q, k, v, coords, codes, GeoShift{}, w_rpe, alpha, out_weight, out_bias, N, H, D,
                                C, K, T, t0, Tl, B, precision, head_groups, transport, workspace, workspace_bytes, xbuf,
                                xbuf_bytes, out_full, stream);
}

extern "C" int hept_forward_sharded_src(hept_comm* comm, const float* q, const float* k, const float* v,
                                        const float* coords, const float* eta_idx, const float* phi_idx,
                                        const float* cfac, int raw_size, const float* w_rpe, const float* alpha,
                                        const float* out_weight, const float* out_bias, int N, int H, int D, int C,
                                        int K, int T, int t0, int Tl, int B, int precision, int head_groups,
                                        int transport, void* workspace, size_t workspace_bytes, void* xbuf,
                                        size_t xbuf_bytes, float* out_full, void* stream) {
    if (!eta_idx || !phi_idx || !cfac) return HEPT_ERR_ARG;
    if (raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    return forward_sharded_impl(comm, q, k, v, coords, nullptr, GeoShift{eta_idx, phi_idx, cfac, raw_size}, w_rpe,
                                alpha, out_weight, out_bias, N, H, D, C, K, T, t0, Tl, B, precision, head_groups,
                                transport, workspace, workspace_bytes, xbuf, xbuf_bytes, out_full, stream);
}

extern "C" int hept_attn_block_forward(const float* x, const float* coords, const int64_t* codes,
                                       const hept_attn_params* p, int N, int H, int D, int C, int K, int T, int B,
                                       int precision, void* workspace, size_t workspace_bytes, float* y,
                                       void* stream) {
    if (!x || !coords || !codes || !p || !workspace || !y) return HEPT_ERR_ARG;
    if (!p->norm1_w || !p->norm1_b || !p->w_q || !p->w_k || !p->w_v || !p->w_rpe || !p->alpha || !p->out_w ||
        !p->norm2_w || !p->norm2_b || !p->ff1_w || !p->ff1_b || !p->ff2_w || !p->ff2_b)
        return HEPT_ERR_ARG;
    int rc = hept_check_shape(N, H, D, C, T, B);
    if (rc) return rc;
    if (D != 24) return HEPT_ERR_SHAPE;
    const Workspace w = carve(workspace, N, H, C, T, precision);
    if (workspace_bytes < w.bytes) return HEPT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    prof_mark(0, st);
    int32_t* qpos = w.pos;
    int32_t* kpos = w.pos + (size_t)T * H * N;
    // any number of tables: the row builder and the sort walk chunks of HEPT_MAX_TABLES, as run_begin does (the rows
    // are rewritten identically by every chunk)
    for (int c0 = 0; c0 < T; c0 += HEPT_MAX_TABLES) {
        const int tc = T - c0 < HEPT_MAX_TABLES ? T - c0 : HEPT_MAX_TABLES;
        // (K == 0: params->w_rpe is sqrt_w (H, C); K > 0: the weight itself, scale computed in the kernel -- see run_begin)
        void* zptr = nullptr;
        size_t zbytes = 0;
        hept_sort_zero_block(w.sort_ws, N, H, tc, &zptr, &zbytes);   // (see run_begin)
        rc = hept_prep_hash_fused_rpe(x, p->norm1_w, p->norm1_b, p->eps1, p->w_q, p->w_k, p->w_v, coords, p->w_rpe, K,
                                      p->alpha, codes, N, N, H, D, C, T, c0, tc, precision, w.qhat, w.kvhat, w.qproj,
                                      w.kproj, w.minmax, stream, zptr, zbytes);
        if (rc) return rc;
        if (c0 == 0) prof_mark(1, st);
        int32_t* cq = T <= HEPT_MAX_TABLES ? qpos : w.pos_chunk;
        int32_t* ck = cq + (size_t)tc * H * N;
        rc = hept_sort_tables_rows(w.qproj, w.kproj, codes, w.minmax, N, H, T, c0, tc, w.sort_ws, cq, ck, nullptr, stream,
                                   zbytes != 0);
        if (rc) return rc;
        if (cq != qpos) {
            const size_t off = (size_t)c0 * H * N, bytes = (size_t)tc * H * N * 4;
            if (hipMemcpyAsync(qpos + off, cq, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(kpos + off, ck, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return HEPT_ERR_LAUNCH;
        }
    }
    prof_mark(2, st);
    rc = hept_block_attn(w.qhat, w.kvhat, qpos, kpos, N, H, D, T, B, precision, w.part, stream);
    if (rc) return rc;
    prof_mark(3, st);
    rc = hept_combine_ffn(w.part, hept_part_precision(precision, D), T, N, H, D, 0, N, p->out_w, p->out_b, x,
                          p->norm2_w, p->norm2_b, p->eps2, p->ff1_w, p->ff1_b, p->ff2_w, p->ff2_b, y, stream);
    prof_mark(4, st);
    prof_call_done();
    return rc;
}

extern "C" int hept_profile_enable(int mode, int max_calls) {
    if (mode < 0 || mode > 2 || max_calls < 0) return HEPT_ERR_ARG;
    if (g_prof.ev) {
        for (size_t i = 0; i < (size_t)g_prof.max_calls * PROF_SLOTS; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev;
        delete[] g_prof.last;
        delete[] g_prof.mid;
        g_prof.ev = nullptr;
        g_prof.last = nullptr;
        g_prof.mid = nullptr;
    }
    g_prof.mode = mode;
    g_prof.n_calls = 0;
    g_prof.seen = 0;
    g_prof.max_calls = mode ? max_calls : 0;
    if (g_prof.max_calls) {
        g_prof.ev = new hipEvent_t[(size_t)g_prof.max_calls * PROF_SLOTS];
        g_prof.last = new int[g_prof.max_calls]();
        g_prof.mid = new int[g_prof.max_calls]();
        for (size_t i = 0; i < (size_t)g_prof.max_calls * PROF_SLOTS; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return HEPT_ERR_LAUNCH;
    }
    return HEPT_OK;
}

extern "C" int hept_profile_stride(int stride) {
    if (stride < 1) return HEPT_ERR_ARG;
    g_prof.stride = stride;
    g_prof.seen = 0;
    return HEPT_OK;
}

extern "C" int hept_profile_read(float* ms, int* n_calls) {
    if (!ms || !n_calls) return HEPT_ERR_ARG;
    for (int i = 0; i < 7; ++i) ms[i] = 0.f;
    *n_calls = g_prof.n_calls;
    for (int c = 0; c < g_prof.n_calls; ++c) {
        hipEvent_t* e = g_prof.ev + (size_t)c * PROF_SLOTS;
        const int first = g_prof.mode == 1 ? 2 : 0, last = g_prof.mode == 1 ? 3 : g_prof.last[c];
        if (last <= first) continue;
        if (hipEventSynchronize(e[last]) != hipSuccess) return HEPT_ERR_LAUNCH;
        for (int sidx = first; sidx < last; ++sidx) {
            float dt = 0.f;
            if (hipEventElapsedTime(&dt, e[sidx], e[sidx + 1]) != hipSuccess) return HEPT_ERR_LAUNCH;
            ms[sidx] += dt;
        }
        if (g_prof.mode == 2 && g_prof.mid[c] && last >= 2) {   // first launch of the sort: end of the row builder -> the mark
            float dt = 0.f;
            if (hipEventElapsedTime(&dt, e[1], e[PROF_SORT_MID]) != hipSuccess) return HEPT_ERR_LAUNCH;
            ms[6] += dt;
        }
    }
    g_prof.n_calls = 0;
    return HEPT_OK;
}
