// block_attn_bwd: backward of the block-local RBF attention, fp32 tiles, fp32 MFMA.
//
// The reference has no custom backward: it trains through example/hept.py:55-80 with plain autograd
// (example/trainer.py:11-22).  Hashing and sorting carry no gradient (lsh_mapping is @torch.no_grad,
// example/hept_utils.py:64; argsort yields integers), so given the forward's permutations the
// differentiable part is, per (table, head, block):
//     S_ij = q^_i.k^_j - |q^_i|^2/2 - |k^_j|^2/2,   P = exp(min(S, 0)),   numer = P.V,   den = P.1
// With G_i = [d numer_i | d den_i] (the upstream gradient row of query i; the same for every table,
// since the tables are summed before the division) and V carrying its 1.0 column:
//     dP = G.V^T                       (the 1.0 column of V adds d den_i to every dP_ij)
//     dS = dP o P o [S <= 0]           (clamp(max=0) passes the gradient where S <= 0, like torch)
//     dV_j  = sum_i P_ij dnumer_i
//     dq^_i = sum_j dS_ij (k^_j - q^_i),     dk^_j = sum_i dS_ij (q^_i - k^_j)
// The scores are recomputed (never stored).  Same MFMA scheme as the forward: the accumulator of
// (rows . cols^T) already has the A-operand layout of the next product, so nothing bounces through LDS.
//   phase Q: wave owns 32 queries, walks the key tiles:   X = K^.Q^T, Y = V.G^T, dS = Y o P o M,  Z += dS^T K^
//   phase K: wave owns 32 keys,    walks the query tiles: X = Q^.K^T, Y = G.V^T, dS likewise,     ZK += dS^T Q^, ZV += P^T G
// The row sums (sum_j dS_ij, sum_i dS_ij) ride in the products through a 1.0 placed in the unused column 30
// of the staged K^ / Q^ tiles.  Outputs are per-table partial rows scattered back to point order:
//   dq_part (T, N, H, 32): d q^ (E columns)          dkv_part (T, N, H, 64): d k^ (E columns) | d v (D columns)
#include "common.h"

namespace {

__device__ __forceinline__ int sw_byte(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ float lds_elem(const char* tile, int row, int col) {
    return *reinterpret_cast<const float*>(tile + sw_byte(row, col >> 2) + ((col & 3) << 2));
}
// 16 consecutive columns [16*hh, 16*hh+16) of a staged row
__device__ __forceinline__ void lds_half_row(const char* tile, int row, int hh, float (&x)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + sw_byte(row, 4 * hh + c));
        x[4 * c] = v[0]; x[4 * c + 1] = v[1]; x[4 * c + 2] = v[2]; x[4 * c + 3] = v[3];
    }
}

template <int NKT, bool FULL>
__global__ __launch_bounds__(64 * NKT) void block_attn_bwd_kernel(
    const float* __restrict__ qhat, const float* __restrict__ kvhat, const int* __restrict__ qpos,
    const int* __restrict__ kpos, const float* __restrict__ gacc, float* __restrict__ dq_part,
    float* __restrict__ dkv_part, int N, int H, int D, int B, int nb) {
    constexpr int NT = 64 * NKT, ROWS = 32 * NKT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_s = smem;                      // [ROWS][32 f32] swizzled, column 30 = 1.0, column 31 = 0
    char* k_s = smem + ROWS * 128;         // same for the keys
    char* v_s = smem + 2 * ROWS * 128;     // [v | 1.0 at column D | 0]
    char* g_s = smem + 3 * ROWS * 128;     // upstream rows [d numer | d den | 0]
    float* qn_s = reinterpret_cast<float*>(smem + 4 * ROWS * 128);
    float* kn_s = qn_s + ROWS;
    int* qidx_s = reinterpret_cast<int*>(kn_s + ROWS);
    int* kidx_s = qidx_s + ROWS;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int bid = blockIdx.x;
    // the tables of a block run side by side (t fastest), as in the forward: the gathers of a point's rows meet in the L2
#ifdef HEPT_BWD_TABLE_MAJOR
    const int h = bid % H, rest = bid / H, b = rest % nb, t = rest / nb;
#else
    const int tl_ = (int)gridDim.x / (H * nb);
    const int h = bid % H, rest = bid / H, b = rest / tl_, t = rest % tl_;
#endif
    const size_t seg = ((size_t)t * H + h) * N + (size_t)b * B;
    const int* __restrict__ qp = qpos + seg;
    const int* __restrict__ kp = kpos + seg;

    // ---- stage the four tiles (gathered rows, 16 B per lane), norms and indices
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // ROWS * 8 chunks of q^ and of G
        const int ci = it * NT + tid, row = ci >> 3, c = ci & 7;
        const bool ok = FULL || row < B;
        const int src = ok ? qp[row] : 0;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            qv = *reinterpret_cast<const f32x4*>(qhat + ((size_t)h * N + src) * 32 + c * 4);
            gv = *reinterpret_cast<const f32x4*>(gacc + ((size_t)src * H + h) * 32 + c * 4);
        }
        if (c == 7) {
            qn_s[row] = qv[3];
            qidx_s[row] = ok ? src : -1;
            qv[3] = 0.f;
            qv[2] = ok ? 1.f : 0.f;  // column 30: row sums of dS ride in the MFMA
        }
        *reinterpret_cast<f32x4*>(q_s + sw_byte(row, c)) = qv;
        *reinterpret_cast<f32x4*>(g_s + sw_byte(row, c)) = gv;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // ROWS * 16 chunks of the kvhat rows
        const int ci = it * NT + tid, row = ci >> 4, c = ci & 15;
        const bool ok = FULL || row < B;
        const int src = ok ? kp[row] : 0;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (ok) x = *reinterpret_cast<const f32x4*>(kvhat + ((size_t)h * N + src) * 64 + c * 4);
        if (c == 7) {
            kn_s[row] = x[3];
            kidx_s[row] = ok ? src : -1;
            x[3] = 0.f;
            x[2] = ok ? 1.f : 0.f;
        }
        if (c < 8)
            *reinterpret_cast<f32x4*>(k_s + sw_byte(row, c)) = x;
        else
            *reinterpret_cast<f32x4*>(v_s + sw_byte(row, c - 8)) = x;
    }
    __syncthreads();

    const int own = w * 32 + li;  // the query (phase Q) / key (phase K) of this lane
    const bool own_ok = FULL || own < B;

    // =========================== phase Q: d q^ ===========================
    {
        float qreg[16], greg[16];
        lds_half_row(q_s, own, hh, qreg);
        lds_half_row(g_s, own, hh, greg);
        if (hh == 1) qreg[14] = 0.f;  // the 1.0 of column 30 is not a feature
        const float qn = qn_s[own];
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (!FULL && kt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = qn + kn_s[kt * 32 + hept_acc_row(r, hh)];
                y[r] = 0.f;
            }
            float kf[16], vf[16];
            lds_half_row(k_s, kt * 32 + li, hh, kf);
            lds_half_row(v_s, kt * 32 + li, hh, vf);
#pragma unroll
            for (int s = 0; s < 16; ++s) x = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qreg[s], x, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) y = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[s], greg[s], y, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + hept_acc_row(r, hh);
                float ds = x[r] <= 0.f ? fminf(__expf(x[r]), 1.f) * y[r] : 0.f;
                if (!FULL && (key >= B || !own_ok)) ds = 0.f;
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, lds_elem(k_s, key, li), z, 0, 0, 0);
            }
        }
        // d q^_i = sum_j dS_ij k^_j - (sum_j dS_ij) q^_i ;  lane = column, registers = queries
        float* __restrict__ dst = dq_part + (size_t)t * N * H * 32 + (size_t)h * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q2 = w * 32 + hept_acc_row(r, hh);
            const float rs = __shfl(z[r], 30 + 32 * hh);
            if (FULL || q2 < B) dst[(size_t)qidx_s[q2] * H * 32] = z[r] - rs * lds_elem(q_s, q2, li);
        }
    }

    // =========================== phase K: d k^, d v ===========================
    {
        float kreg[16], vreg[16];
        lds_half_row(k_s, own, hh, kreg);
        lds_half_row(v_s, own, hh, vreg);
        if (hh == 1) kreg[14] = 0.f;
        const float kn = kn_s[own];
        f32x16 zk, zv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { zk[r] = 0.f; zv[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt) {
            if (!FULL && qt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = qn_s[qt * 32 + hept_acc_row(r, hh)] + kn;
                y[r] = 0.f;
            }
            float qf[16], gf[16];
            lds_half_row(q_s, qt * 32 + li, hh, qf);
            lds_half_row(g_s, qt * 32 + li, hh, gf);
#pragma unroll
            for (int s = 0; s < 16; ++s) x = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s], kreg[s], x, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) y = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[s], vreg[s], y, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qry = qt * 32 + hept_acc_row(r, hh);
                float p = fminf(__expf(x[r]), 1.f);
                if (!FULL && (qry >= B || !own_ok)) p = 0.f;
                const float ds = x[r] <= 0.f ? p * y[r] : 0.f;
                zk = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, lds_elem(q_s, qry, li), zk, 0, 0, 0);
                zv = __builtin_amdgcn_mfma_f32_32x32x2f32(p, lds_elem(g_s, qry, li), zv, 0, 0, 0);
            }
        }
        float* __restrict__ dst = dkv_part + (size_t)t * N * H * 64 + (size_t)h * 64 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = w * 32 + hept_acc_row(r, hh);
            const float rs = __shfl(zk[r], 30 + 32 * hh);
            if (FULL || k2 < B) {
                float* row = dst + (size_t)kidx_s[k2] * H * 64;
                row[0] = zk[r] - rs * lds_elem(k_s, k2, li);
                row[32] = li < D ? zv[r] : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same backward on the bf16 matrix pipe (the default; the f32-MFMA kernel above is kept as ground truth).
//
// Every f32 factor is split into bf16 pieces (common.h) and each f32 product becomes 6 (or 5) bf16 MFMAs with f32
// accumulation -- 74 bf16 MFMAs of 8 passes per 32x32 tile pair instead of 112 f32 MFMAs of 16 passes.  Which
// products need all the pieces: the logits X (exponentiated), dP = G.V^T (dnumer.v + dden cancels where the
// block's values agree with the output) and the sums dS.K^, dS.Q^ (they cancel against rowsum(dS).q^: the row sum
// rides through the same MFMAs against a 1.0 column, so the error of the two-piece dS is common to both terms
// and only the three-piece K^/Q^ matter).  One workgroup, 6 planes of LDS (48 KB at B = 128), two phases:
//   stage K^ | V planes, own Q^ / G rows in registers         -> phase Q (d q^)
//   own K^ / V rows from LDS to registers, then the waves write their Q^ / G rows over the planes -> phase K
// Column 30 / 31 bookkeeping: q^ rows carry (-|q|^2/2, 1), k^ rows carry (1, -|k|^2/2) there, so the logit needs no
// separate norm terms; rowsum(dS) comes out in column 30 of dS.K^ and in column 31 of dS^T.Q^.
constexpr int BPROW = 64;  // bytes of one 32-column bf16 plane row
__device__ __forceinline__ int plane_chunk(int row, int c) { return row * BPROW + ((c ^ ((row >> 2) & 3)) << 4); }

// LDS addresses are lane-dependent bases (computed once) plus compile-time offsets: tile bases are multiples of 16
// rows, so the swizzle term (row >> 2) & 3 of a fragment row depends on the lane and on "+8 rows" only.
struct PlaneLanes {
    int row[2];  // A-operand 16-B chunk of row li, step s:      + tile * 32 * BPROW
    int tr[2];   // transposed 8-B reads, rows +0..3 / +8..11:   + (tile * 32 + 16 s) * BPROW
};
__device__ __forceinline__ PlaneLanes plane_lanes(int lane) {
    const int li = lane & 31, hh = lane >> 5, l16 = lane & 15, g = lane >> 4;
    PlaneLanes p;
    p.row[0] = plane_chunk(li, hh);
    p.row[1] = plane_chunk(li, 2 + hh);
    const int r = 4 * hh + (l16 >> 2), colb = 32 * (g & 1) + 8 * (l16 & 3);
    p.tr[0] = plane_chunk(r, colb >> 4) + (colb & 15);
    p.tr[1] = plane_chunk(r + 8, colb >> 4) + (colb & 15);
    return p;
}

// B-operand fragment (k = the 16 rows starting at byte offset base, n = column li) by transposing reads
__device__ __forceinline__ u32x4 plane_tr_frag(const char* plane, const PlaneLanes& pl, int base) {
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + base + pl.tr[0]));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + base + pl.tr[1]));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(u32x4, vv);
}

// x += a.b with a (three planes in LDS, A operand rows) and b (three pieces in registers): the six leading terms
__device__ __forceinline__ f32x16 mfma6(const char* planes, int plane_bytes, int off, const u32x4 (&b)[3], f32x16 x) {
    const u32x4 ah = *reinterpret_cast<const u32x4*>(planes + off);
    const u32x4 am = *reinterpret_cast<const u32x4*>(planes + plane_bytes + off);
    const u32x4 al = *reinterpret_cast<const u32x4*>(planes + 2 * plane_bytes + off);
    x = mfma_bf16(al, b[0], x);
    x = mfma_bf16(ah, b[2], x);
    x = mfma_bf16(am, b[1], x);
    x = mfma_bf16(am, b[0], x);
    x = mfma_bf16(ah, b[1], x);
    x = mfma_bf16(ah, b[0], x);
    return x;
}
// z += a.b with a (two pieces in registers, accumulator layout) and b (three planes, transposed reads): five terms
__device__ __forceinline__ f32x16 mfma5_tr(const u32x4& ah, const u32x4& am, const char* planes, int plane_bytes,
                                           const PlaneLanes& pl, int base, f32x16 z) {
    const u32x4 bh = plane_tr_frag(planes, pl, base);
    const u32x4 bm = plane_tr_frag(planes + plane_bytes, pl, base);
    const u32x4 bl = plane_tr_frag(planes + 2 * plane_bytes, pl, base);
    z = mfma_bf16(ah, bl, z);
    z = mfma_bf16(am, bm, z);
    z = mfma_bf16(am, bh, z);
    z = mfma_bf16(ah, bm, z);
    z = mfma_bf16(ah, bh, z);
    return z;
}

// Element (row, column li) of an f32 tile held as three bf16 planes (the sum of its pieces), for the rows a lane
// meets in the accumulator layout: row = 32 w + acc_row(r, hh).  Its byte offset is a lane constant plus a
// compile-time term: the swizzle of such a row is hh ^ (2 if r & 4), i.e. bit 5 of the offset flips for r & 4.
struct ElemLanes {
    int base[2];  // r & 4 == 0 / != 0
};
__device__ __forceinline__ ElemLanes elem_lanes(int w, int lane) {
    const int li = lane & 31, hh = lane >> 5;
    ElemLanes e;
    e.base[0] = (w * 32 + 4 * hh) * BPROW + ((((li >> 3) ^ hh) & 3) << 4) + ((li & 7) << 1);
    e.base[1] = e.base[0] ^ 32;
    return e;
}
__device__ __forceinline__ float plane_elem3(const char* planes, int plane_bytes, const ElemLanes& e, int r) {
    const int off = e.base[(r >> 2) & 1] + ((r & 3) + 8 * (r >> 2)) * BPROW;
    float v = 0.f;
#pragma unroll
    for (int pc = 2; pc >= 0; --pc)
        v += __uint_as_float((unsigned int)*reinterpret_cast<const unsigned short*>(planes + pc * plane_bytes + off) << 16);
    return v;
}

template <int NKT, bool FULL>
__global__ __launch_bounds__(64 * NKT) __attribute__((amdgpu_waves_per_eu((NKT <= 4 && FULL) ? 3 : 2, (NKT <= 4 && FULL) ? 3 : 2)))
void block_attn_bwd_split_kernel(
    const float* __restrict__ qhat, const float* __restrict__ kvhat, const int* __restrict__ qpos,
    const int* __restrict__ kpos, const float* __restrict__ gacc, float* __restrict__ dq_part,
    float* __restrict__ dkv_part, int N, int H, int D, int B, int nb) {
    constexpr int NT = 64 * NKT, ROWS = 32 * NKT, PL = ROWS * BPROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_s = smem;           // phase Q: K^ planes (h, m, l)   phase K: Q^ planes
    char* b_s = smem + 3 * PL;  // phase Q: V  planes             phase K: G  planes

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int bid = blockIdx.x;
    // the tables of a block run side by side (t fastest), as in the forward: the gathers of a point's rows meet in the L2
#ifdef HEPT_BWD_TABLE_MAJOR
    const int h = bid % H, rest = bid / H, b = rest % nb, t = rest / nb;
#else
    const int tl_ = (int)gridDim.x / (H * nb);
    const int h = bid % H, rest = bid / H, b = rest / tl_, t = rest % tl_;
#endif
    const size_t seg = ((size_t)t * H + h) * N + (size_t)b * B;
    const int* __restrict__ qp = qpos + seg;
    const int* __restrict__ kp = kpos + seg;
    const float* __restrict__ qbase = qhat + (size_t)h * N * 32;
    const float* __restrict__ kvbase = kvhat + (size_t)h * N * 64;

    const int own = w * 32 + li;  // the query (phase Q) / key (phase K) of this lane
    const bool own_ok = FULL || own < B;
    const int qsrc = qp[own_ok ? own : 0], ksrc = kp[own_ok ? own : 0];
    const PlaneLanes pl = plane_lanes(lane);

    // ---- own Q^ and G rows -> bf16 pieces in registers (B-operand layout: step s, half hh = columns 16 s + 8 hh ..)
    u32x4 q3[2][3], g3[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (own_ok) {
            const float* qrow = qbase + (size_t)qsrc * 32 + 16 * s + 8 * hh;
            const float* grow = gacc + ((size_t)qsrc * H + h) * 32 + 16 * s + 8 * hh;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(qrow), a1 = *reinterpret_cast<const f32x4*>(qrow + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(grow), g1 = *reinterpret_cast<const f32x4*>(grow + 4);
            a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
            gg[0] = g0[0]; gg[1] = g0[1]; gg[2] = g0[2]; gg[3] = g0[3];
            gg[4] = g1[0]; gg[5] = g1[1]; gg[6] = g1[2]; gg[7] = g1[3];
            if (s == 1 && hh == 1) {  // columns (30, 31) = (-|q|^2/2, 1)
                a[6] = a[7];
                a[7] = 1.f;
            }
        }
        split3_bf16(a, q3[s][0], q3[s][1], q3[s][2]);
        split3_bf16(gg, g3[s][0], g3[s][1], g3[s][2]);
    }

    // ---- stage K^ and V planes: one item = 8 consecutive columns of a gathered kvhat row
    for (int ci = tid; ci < ROWS * 8; ci += NT) {
        const int key = ci >> 3, c = ci & 7;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (FULL || key < B) {
            const float* src = kvbase + (size_t)kp[key] * 64 + c * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
            a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
            if (c == 3) a[6] = 1.f;  // columns (30, 31) = (1, -|k|^2/2)
        }
        u32x4 ph, pm, pl;
        split3_bf16(a, ph, pm, pl);
        char* dst = (c < 4 ? a_s : b_s) + plane_chunk(key, c & 3);
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + PL) = pm;
        *reinterpret_cast<u32x4*>(dst + 2 * PL) = pl;
    }
    __syncthreads();

    // =========================== phase Q: d q^ ===========================
    f32x16 z;
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (!FULL && kt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = 0.f; y[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = kt * 32 * BPROW + pl.row[s];
                x = mfma6(a_s, PL, off, q3[s], x);  // X^T = K^ . Q^T (+ both norms)
                y = mfma6(b_s, PL, off, g3[s], y);  // Y^T = V . G^T
            }
            float ds[16], pe[16];
            exp_clamped(x, pe);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ds[r] = x[r] <= 0.f ? pe[r] * y[r] : 0.f;  // exp(x) <= 1 wherever the clamp passes the gradient
                if (!FULL && (kt * 32 + hept_acc_row(r, hh) >= B || !own_ok)) ds[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float pa[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[j] = ds[8 * s + j];
                u32x4 dh, dm;
                split2_bf16(pa, dh, dm);
                z = mfma5_tr(dh, dm, a_s, PL, pl, (kt * 32 + 16 * s) * BPROW, z);  // Z += dS . K^ ; column 30 = rowsum(dS)
            }
        }
    }

    // ---- hand-over between the phases.  After the barrier nobody reads the K^ / V planes as tiles any more and
    // every wave touches only its own 32 rows of each plane until the next barrier:
    //   own V rows -> registers;  own Q^ pieces -> the V planes (rows now private);  the phase Q epilogue reads q^ from
    //   there with lane = column (reading it from global memory instead -- 16 dependent row reads per wave and
    //   phase -- cost 15 % of the kernel);  own K^ rows -> registers;  Q^ / G pieces -> the planes of phase K.
    u32x4 k3[2][3], v3[2][3];
    const ElemLanes el = elem_lanes(w, lane);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int off = w * 32 * BPROW + pl.row[s];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) v3[s][pc] = *reinterpret_cast<const u32x4*>(b_s + pc * PL + off);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(b_s + pc * PL + off) = q3[s][pc];
    }
    {   // d q^_i = sum_j dS_ij k^_j - (sum_j dS_ij) q^_i ;  lane = column, registers = queries
        float* __restrict__ dst = dq_part + (size_t)t * N * H * 32 + (size_t)h * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q2 = w * 32 + hept_acc_row(r, hh);
            const int src = __shfl(qsrc, hept_acc_row(r, hh));
            const float rs = __shfl(z[r], 30 + 32 * hh);
            const float qv = plane_elem3(b_s, PL, el, r);
            if (FULL || q2 < B) hept_st<HEPT_NT_BWD_ROWS>(dst + (size_t)src * H * 32, li < 30 ? z[r] - rs * qv : 0.f);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int off = w * 32 * BPROW + pl.row[s];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            k3[s][pc] = *reinterpret_cast<const u32x4*>(a_s + pc * PL + off);
            *reinterpret_cast<u32x4*>(a_s + pc * PL + off) = q3[s][pc];
            *reinterpret_cast<u32x4*>(b_s + pc * PL + off) = g3[s][pc];
        }
    }
    __syncthreads();

    // =========================== phase K: d k^, d v ===========================
    {
        f32x16 zk, zv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { zk[r] = 0.f; zv[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt) {
            if (!FULL && qt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = 0.f; y[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = qt * 32 * BPROW + pl.row[s];
                x = mfma6(a_s, PL, off, k3[s], x);  // X = Q^ . K^T (+ both norms)
                y = mfma6(b_s, PL, off, v3[s], y);  // Y = G . V^T
            }
            float pr[16], ds[16];
            exp_clamped(x, pr);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!FULL && (qt * 32 + hept_acc_row(r, hh) >= B || !own_ok)) pr[r] = 0.f;
                ds[r] = x[r] <= 0.f ? pr[r] * y[r] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float pa[8], da[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { pa[j] = pr[8 * s + j]; da[j] = ds[8 * s + j]; }
                u32x4 ph, pm, dh, dm;
                split2_bf16(pa, ph, pm);
                split2_bf16(da, dh, dm);
                zk = mfma5_tr(dh, dm, a_s, PL, pl, (qt * 32 + 16 * s) * BPROW, zk);  // ZK += dS^T . Q^ ; column 31 = colsum(dS)
                zv = mfma5_tr(ph, pm, b_s, PL, pl, (qt * 32 + 16 * s) * BPROW, zv);  // ZV += P^T . G
            }
        }
        // the wave's K^ pieces go back into (its own rows of) the first three planes so that the epilogue can read
        // k^ with lane = column
        __syncthreads();  // every wave is done with the Q^ planes
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = w * 32 * BPROW + pl.row[s];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(a_s + pc * PL + off) = k3[s][pc];
        }
        float* __restrict__ dst = dkv_part + (size_t)t * N * H * 64 + (size_t)h * 64 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = w * 32 + hept_acc_row(r, hh);
            const int src = __shfl(ksrc, hept_acc_row(r, hh));
            const float rs = __shfl(zk[r], 31 + 32 * hh);
            const float kv = plane_elem3(a_s, PL, el, r);
            if (FULL || k2 < B) {
                float* row = dst + (size_t)src * H * 64;
                hept_st<HEPT_NT_BWD_ROWS>(row, li < 30 ? zk[r] - rs * kv : 0.f);
                hept_st<HEPT_NT_BWD_ROWS>(row + 32, li < D ? zv[r] : 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-bit tiles (HEPT_PREC_BF16 training, opt-in): the backward on the rows the bf16 forward gathers -- q^ rows of
// 64 B, k^ | v rows of 128 B, the f32 norm in the last four bytes of a q^ / k^ row -- with ONE bf16 MFMA per product
// where the kernel above issues five or six.  Same two phases and the same operand layouts as the split kernel on
// single planes (PlaneLanes / plane_tr_frag / ElemLanes above), but all four tiles (Q^, K^, V, G) are resident at
// once (32 KB at B = 128: no hand-over between the phases), the norms are added to the logits in f32 as the bf16
// forward does (a norm rounded to bf16 would be off by 2^-9 |q|^2), and columns 30 / 31 of the staged tiles carry only
// the 1.0 that makes the products return the row / column sums of dS: K^ (1, 0), Q^ (0, 1) -- their cross terms in
// Q^.K^ vanish.  G (the upstream rows, f32) and the tile-level factors P, dS are rounded to bf16; every accumulation
// is f32.  Gradients agree with the f32-tile backward to the accuracy of a bf16 forward (tests/test_gpu_backward.py).
__device__ __forceinline__ float plane_elem1(const char* plane, const ElemLanes& e, int r) {
    const int off = e.base[(r >> 2) & 1] + ((r & 3) + 8 * (r >> 2)) * BPROW;
    return __uint_as_float((unsigned int)*reinterpret_cast<const unsigned short*>(plane + off) << 16);
}
__device__ __forceinline__ u32x4 pack8_bf16(const float* a) {
    return u32x4{hept_pack_bf16(a[0], a[1]), hept_pack_bf16(a[2], a[3]), hept_pack_bf16(a[4], a[5]),
                 hept_pack_bf16(a[6], a[7])};
}

// (The per-table gradient rows leave the kernel as bf16 too -- dq_part (T, N, H, 32) and dkv_part (T, N, H, 64) in
//  bf16, two columns per dword: with f32 rows the kernel was bound by writing them, 553 MB at tracking-60k, 260 us, and
//  the table sum by reading them back; the sum itself (hept_bwd_reduce16) is f32.)
template <int NKT, bool FULL>
__global__ __launch_bounds__(64 * NKT) void block_attn_bwd_bf16_kernel(
    const char* __restrict__ qhat, const char* __restrict__ kvhat, const int* __restrict__ qpos,
    const int* __restrict__ kpos, const float* __restrict__ gacc, unsigned int* __restrict__ dq_part,
    unsigned int* __restrict__ dkv_part, int N, int H, int D, int B, int nb) {
    constexpr int NT = 64 * NKT, ROWS = 32 * NKT, PL = ROWS * BPROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_s = smem;              // Q^ rows, columns (30, 31) = (0, 1.0)
    char* k_s = smem + PL;         // K^ rows, columns (30, 31) = (1.0, 0)
    char* v_s = smem + 2 * PL;     // [v | 1.0 at column D | 0]
    char* g_s = smem + 3 * PL;     // upstream rows [d numer | d den | 0], rounded to bf16
    float* qn_s = reinterpret_cast<float*>(smem + 4 * PL);
    float* kn_s = qn_s + ROWS;
    int* qidx_s = reinterpret_cast<int*>(kn_s + ROWS);
    int* kidx_s = qidx_s + ROWS;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hh = lane >> 5, li = lane & 31;
    const int bid = blockIdx.x;
    // the tables of a block run side by side (t fastest), as in the forward: the gathers of a point's rows meet in the L2
#ifdef HEPT_BWD_TABLE_MAJOR
    const int h = bid % H, rest = bid / H, b = rest % nb, t = rest / nb;
#else
    const int tl_ = (int)gridDim.x / (H * nb);
    const int h = bid % H, rest = bid / H, b = rest / tl_, t = rest % tl_;
#endif
    const size_t seg = ((size_t)t * H + h) * N + (size_t)b * B;
    const int* __restrict__ qp = qpos + seg;
    const int* __restrict__ kp = kpos + seg;

    // ---- stage the four tiles: one item = a 16-B chunk (8 columns) of a gathered row
    for (int ci = tid; ci < ROWS * 4; ci += NT) {
        const int row = ci >> 2, c = ci & 3;
        const bool ok = FULL || row < B;
        const int src = ok ? qp[row] : 0;
        u32x4 qv = {0u, 0u, 0u, 0u}, gv = {0u, 0u, 0u, 0u};
        if (ok) {
            qv = *reinterpret_cast<const u32x4*>(qhat + ((size_t)h * N + src) * 64 + c * 16);
            const float* grow = gacc + ((size_t)src * H + h) * 32 + c * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(grow), g1 = *reinterpret_cast<const f32x4*>(grow + 4);
            gv = u32x4{hept_pack_bf16(g0[0], g0[1]), hept_pack_bf16(g0[2], g0[3]), hept_pack_bf16(g1[0], g1[1]),
                       hept_pack_bf16(g1[2], g1[3])};
        }
        if (c == 3) {
            qn_s[row] = __uint_as_float(qv[3]);
            qidx_s[row] = ok ? src : -1;
            qv[3] = ok ? 0x3F800000u : 0u;   // columns (30, 31) = (0, 1.0): column sums of dS ride in dS^T . Q^
        }
        *reinterpret_cast<u32x4*>(q_s + plane_chunk(row, c)) = qv;
        *reinterpret_cast<u32x4*>(g_s + plane_chunk(row, c)) = gv;
    }
    for (int ci = tid; ci < ROWS * 8; ci += NT) {
        const int row = ci >> 3, c = ci & 7;
        const bool ok = FULL || row < B;
        const int src = ok ? kp[row] : 0;
        u32x4 x = {0u, 0u, 0u, 0u};
        if (ok) x = *reinterpret_cast<const u32x4*>(kvhat + ((size_t)h * N + src) * 128 + c * 16);
        if (c == 3) {
            kn_s[row] = __uint_as_float(x[3]);
            kidx_s[row] = ok ? src : -1;
            x[3] = ok ? 0x00003F80u : 0u;    // columns (30, 31) = (1.0, 0): row sums of dS ride in dS . K^
        }
        *reinterpret_cast<u32x4*>((c < 4 ? k_s : v_s) + plane_chunk(row, c & 3)) = x;
    }
    __syncthreads();

    const int own = w * 32 + li;  // the query (phase Q) / key (phase K) of this lane
    const bool own_ok = FULL || own < B;
    const PlaneLanes pl = plane_lanes(lane);
    const ElemLanes el = elem_lanes(w, lane);

    // =========================== phase Q: d q^ ===========================
    {
        u32x4 qown[2], gown[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qown[s] = *reinterpret_cast<const u32x4*>(q_s + w * 32 * BPROW + pl.row[s]);
            gown[s] = *reinterpret_cast<const u32x4*>(g_s + w * 32 * BPROW + pl.row[s]);
        }
        const float qn = qn_s[own];
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (!FULL && kt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = qn + kn_s[kt * 32 + hept_acc_row(r, hh)];
                y[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = kt * 32 * BPROW + pl.row[s];
                x = mfma_bf16(*reinterpret_cast<const u32x4*>(k_s + off), qown[s], x);  // X^T = K^ . Q^T (+ norms above)
                y = mfma_bf16(*reinterpret_cast<const u32x4*>(v_s + off), gown[s], y);  // Y^T = V . G^T
            }
            float ds[16], pe[16];
            exp_clamped(x, pe);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (no [X <= 0] mask here: X = -|q_r - k_r|^2 / 2 <= 0 in exact arithmetic, a positive X is f32 round-off
                //  around zero where the exact logit is negative and the gradient does flow; the f32-tile kernels keep
                //  the mask to follow torch's clamp on the same rounded number.  2 of ~10 VALU instructions per logit.)
                ds[r] = pe[r] * y[r];
                if (!FULL && (kt * 32 + hept_acc_row(r, hh) >= B || !own_ok)) ds[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)   // Z += dS . K^ ; column 30 = rowsum(dS)
                z = mfma_bf16(pack8_bf16(ds + 8 * s), plane_tr_frag(k_s, pl, (kt * 32 + 16 * s) * BPROW), z);
        }
        unsigned int* __restrict__ dst = dq_part + (size_t)t * N * H * 16 + (size_t)h * 16 + (li >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q2 = w * 32 + hept_acc_row(r, hh);
            const float rs = __shfl(z[r], 30 + 32 * hh);
            const float qv = plane_elem1(q_s, el, r);
            const float mine = li < 30 ? z[r] - rs * qv : 0.f;
            const float nbr = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mine), 0xB1, 0xF,
                                                                                 0xF, true));  // lane ^ 1
            if ((FULL || q2 < B) && (li & 1) == 0) dst[(size_t)qidx_s[q2] * H * 16] = hept_pack_bf16(mine, nbr);
        }
    }

    // =========================== phase K: d k^, d v ===========================
    {
        u32x4 kown[2], vown[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kown[s] = *reinterpret_cast<const u32x4*>(k_s + w * 32 * BPROW + pl.row[s]);
            vown[s] = *reinterpret_cast<const u32x4*>(v_s + w * 32 * BPROW + pl.row[s]);
        }
        const float kn = kn_s[own];
        f32x16 zk, zv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { zk[r] = 0.f; zv[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NKT; ++qt) {
            if (!FULL && qt * 32 >= B) break;
            f32x16 x, y;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = qn_s[qt * 32 + hept_acc_row(r, hh)] + kn;
                y[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = qt * 32 * BPROW + pl.row[s];
                x = mfma_bf16(*reinterpret_cast<const u32x4*>(q_s + off), kown[s], x);  // X = Q^ . K^T
                y = mfma_bf16(*reinterpret_cast<const u32x4*>(g_s + off), vown[s], y);  // Y = G . V^T
            }
            float pr[16], ds[16];
            exp_clamped(x, pr);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!FULL && (qt * 32 + hept_acc_row(r, hh) >= B || !own_ok)) pr[r] = 0.f;
                ds[r] = pr[r] * y[r];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int base = (qt * 32 + 16 * s) * BPROW;
                zk = mfma_bf16(pack8_bf16(ds + 8 * s), plane_tr_frag(q_s, pl, base), zk);  // dS^T . Q^ ; column 31 = colsum(dS)
                zv = mfma_bf16(pack8_bf16(pr + 8 * s), plane_tr_frag(g_s, pl, base), zv);  // P^T . G
            }
        }
        unsigned int* __restrict__ dst = dkv_part + (size_t)t * N * H * 32 + (size_t)h * 32 + (li >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = w * 32 + hept_acc_row(r, hh);
            const float cs = __shfl(zk[r], 31 + 32 * hh);
            const float kv = plane_elem1(k_s, el, r);
            const float dkm = li < 30 ? zk[r] - cs * kv : 0.f, dvm = li < D ? zv[r] : 0.f;
            const float dkn = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dkm), 0xB1, 0xF,
                                                                                 0xF, true));
            const float dvn = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dvm), 0xB1, 0xF,
                                                                                 0xF, true));
            if ((FULL || k2 < B) && (li & 1) == 0) {
                unsigned int* row = dst + (size_t)kidx_s[k2] * H * 32;
                row[0] = hept_pack_bf16(dkm, dkn);
                row[16] = hept_pack_bf16(dvm, dvn);
            }
        }
    }
}

// sum the per-table partial rows and undo the augmentation:
//   dq,dk,dv (N, H*D);  dcs (N, H, C) = gradient of the scaled coordinates sqrt_w[h,c]*coords[n,c] (q^ and k^ share them)
// One thread per (point, head, column): nine independent loads each, 60 000 workgroups -- the kernel runs at the
// copy bandwidth (a variant that walked 16 points per workgroup to keep d_sqrt_w sums in registers ran at 60 % of it).
// the same sum over bf16 partial rows (the 16-bit training tiles); f32 accumulation in table order.  One thread per
// (point, head, column PAIR): dword loads (one element per thread was 2-byte loads: 3.2 TB/s).
__global__ __launch_bounds__(256) void bwd_reduce16_kernel(const unsigned int* __restrict__ dq_part,
                                                           const unsigned int* __restrict__ dkv_part, int Tl, int N,
                                                           int H, int D, int C, int raw_size, float* __restrict__ dq,
                                                           float* __restrict__ dk, float* __restrict__ dv,
                                                           float* __restrict__ dcs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n, h, column pair)
    const size_t total = (size_t)N * H * 16;
    if (i >= total) return;
    const int col = 2 * (int)(i & 15);
    const size_t nh = i >> 4;  // n * H + h
    float sq[2] = {0.f, 0.f}, sk[2] = {0.f, 0.f}, sv[2] = {0.f, 0.f};
    for (int t = 0; t < Tl; ++t) {
        const unsigned int a = dq_part[(size_t)t * total + i];
        const unsigned int b = dkv_part[((size_t)t * N * H + nh) * 32 + (col >> 1)];
        const unsigned int c = dkv_part[((size_t)t * N * H + nh) * 32 + 16 + (col >> 1)];
        sq[0] += hept_bf16_lo(a); sq[1] += hept_bf16_hi(a);
        sk[0] += hept_bf16_lo(b); sk[1] += hept_bf16_hi(b);
        sv[0] += hept_bf16_lo(c); sv[1] += hept_bf16_hi(c);
    }
    if (nh >= (size_t)raw_size * H) sq[0] = sq[1] = sk[0] = sk[1] = sv[0] = sv[1] = 0.f;  // src variant: padding rows
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int cc = col + e;
        if (cc < D) {
            dq[nh * D + cc] = sq[e];
            dk[nh * D + cc] = sk[e];
            dv[nh * D + cc] = sv[e];
        } else if (cc < D + C) {
            dcs[nh * C + (cc - D)] = sq[e] + sk[e];
        }
    }
}

__global__ __launch_bounds__(256) void bwd_reduce_kernel(const float* __restrict__ dq_part,
                                                         const float* __restrict__ dkv_part, int Tl, int N, int H,
                                                         int D, int C, int raw_size, float* __restrict__ dq,
                                                         float* __restrict__ dk, float* __restrict__ dv,
                                                         float* __restrict__ dcs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n, h, col)
    const size_t total = (size_t)N * H * 32;
    if (i >= total) return;
    const int col = (int)(i & 31);
    const size_t nh = i >> 5;  // n * H + h
    float sq = 0.f, sk = 0.f, sv = 0.f;
    for (int t = 0; t < Tl; ++t) {
        sq += dq_part[(size_t)t * total + i];
        sk += dkv_part[((size_t)t * N * H + nh) * 64 + col];
        sv += dkv_part[((size_t)t * N * H + nh) * 64 + 32 + col];
    }
    if (nh >= (size_t)raw_size * H) sq = sk = sv = 0.f;  // the src variant's zero-filled padding rows
    if (col < D) {
        dq[nh * D + col] = sq;
        dk[nh * D + col] = sk;
        dv[nh * D + col] = sv;
    } else if (col < D + C) {
        dcs[nh * C + (col - D)] = sq + sk;
    }
}

// d_sqrt_w[h][c] = sum_n dcs[n][h][c] * coords[n][c]: lane = (h, c) column of the H*C <= 64 wide rows, a wave walks
// every fourth point of the workgroup's 256 (8 rows in flight), the four waves meet in LDS, and the workgroup's 64
// sums go to partial[wg][64]; dsw_sum_kernel adds the workgroups in index order (a float atomicAdd per column made
// the training gradients depend on the order in which workgroups retire)
constexpr int DSW_POINTS = 256;
__global__ __launch_bounds__(256) void dsw_kernel(const float* __restrict__ dcs, const float* __restrict__ coords, int N,
                                                  int H, int C, float* __restrict__ partial) {
    __shared__ float red_s[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, HC = H * C;
    const int n_begin = blockIdx.x * DSW_POINTS, n_end = min(N, n_begin + DSW_POINTS);
    float acc = 0.f;
    if (lane < HC) {
        const int c = lane % C;
        constexpr int UN = 8;
        for (int n0 = n_begin + w; n0 < n_end; n0 += 4 * UN) {
            float g[UN], x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int n = n0 + 4 * u;
                g[u] = n < n_end ? dcs[(size_t)n * HC + lane] : 0.f;
                x[u] = n < n_end ? coords[(size_t)n * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = fmaf(g[u], x[u], acc);
        }
    }
    red_s[w][lane] = acc;
    __syncthreads();
    if (w == 0) partial[(size_t)blockIdx.x * 64 + lane] = lane < HC ? red_s[0][lane] + red_s[1][lane] + red_s[2][lane] + red_s[3][lane] : 0.f;
}

__global__ __launch_bounds__(256) void dsw_sum_kernel(const float* __restrict__ partial, int n_wgs, int HC,
                                                      float* __restrict__ d_sqrt_w) {
    __shared__ float red_s[HEPT_FSUM_SLICES * HEPT_FSUM_OUT];
    const int o = blockIdx.x * HEPT_FSUM_OUT + threadIdx.x % HEPT_FSUM_OUT;
    const float tot = hept_fixed_sum(partial, n_wgs, 64, o, o < 64, red_s);
    if (threadIdx.x < HEPT_FSUM_OUT && o < HC) d_sqrt_w[o] = tot;
}

template <bool FULL>
int launch_bwd(int nkt, dim3 grid, hipStream_t st, const float* qhat, const float* kvhat, const int* qpos,
               const int* kpos, const float* gacc, float* dq_part, float* dkv_part, int N, int H, int D, int B,
               int nb) {
#define HEPT_BWD_CASE(K)                                                                                         \
    case K: {                                                                                                    \
        constexpr size_t lds = (size_t)4 * 32 * K * 128 + 32 * K * 16;                                           \
        static LdsRaised raised;                                                                                 \
        if (lds > 65536 &&                                                                                       \
            hept_raise_lds(raised, reinterpret_cast<const void*>(&block_attn_bwd_kernel<K, FULL>), lds))         \
            return HEPT_ERR_LAUNCH;                                                                              \
        hipLaunchKernelGGL((block_attn_bwd_kernel<K, FULL>), grid, dim3(64 * K), lds, st, qhat, kvhat, qpos,     \
                           kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);                                       \
        break;                                                                                                   \
    }
    switch (nkt) {
        HEPT_BWD_CASE(1)
        HEPT_BWD_CASE(2)
        HEPT_BWD_CASE(3)
        HEPT_BWD_CASE(4)
        HEPT_BWD_CASE(5)
        HEPT_BWD_CASE(6)
        HEPT_BWD_CASE(7)
        HEPT_BWD_CASE(8)
        default:
            return HEPT_ERR_SHAPE;
    }
#undef HEPT_BWD_CASE
    return hept_launch_status();
}

template <bool FULL>
int launch_bwd_split(int nkt, dim3 grid, hipStream_t st, const float* qhat, const float* kvhat, const int* qpos,
                     const int* kpos, const float* gacc, float* dq_part, float* dkv_part, int N, int H, int D, int B,
                     int nb) {
#define HEPT_BWDS_CASE(K)                                                                                        \
    case K: {                                                                                                    \
        constexpr size_t lds = (size_t)6 * 32 * K * BPROW;                                                       \
        static LdsRaised raised;                                                                                 \
        if (lds > 65536 &&                                                                                       \
            hept_raise_lds(raised, reinterpret_cast<const void*>(&block_attn_bwd_split_kernel<K, FULL>), lds))   \
            return HEPT_ERR_LAUNCH;                                                                              \
        hipLaunchKernelGGL((block_attn_bwd_split_kernel<K, FULL>), grid, dim3(64 * K), lds, st, qhat, kvhat,     \
                           qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);                                 \
        break;                                                                                                   \
    }
    switch (nkt) {
        HEPT_BWDS_CASE(1)
        HEPT_BWDS_CASE(2)
        HEPT_BWDS_CASE(3)
        HEPT_BWDS_CASE(4)
        HEPT_BWDS_CASE(5)
        HEPT_BWDS_CASE(6)
        HEPT_BWDS_CASE(7)
        HEPT_BWDS_CASE(8)
        default:
            return HEPT_ERR_SHAPE;
    }
#undef HEPT_BWDS_CASE
    return hept_launch_status();
}

template <bool FULL>
int launch_bwd_bf16(int nkt, dim3 grid, hipStream_t st, const char* qhat, const char* kvhat, const int* qpos,
                    const int* kpos, const float* gacc, unsigned int* dq_part, unsigned int* dkv_part, int N, int H, int D,
                    int B, int nb) {
#define HEPT_BWDH_CASE(K)                                                                                        \
    case K: {                                                                                                    \
        constexpr size_t lds = (size_t)4 * 32 * K * BPROW + 32 * K * 16;                                         \
        static LdsRaised raised;                                                                                 \
        if (lds > 65536 &&                                                                                       \
            hept_raise_lds(raised, reinterpret_cast<const void*>(&block_attn_bwd_bf16_kernel<K, FULL>), lds))    \
            return HEPT_ERR_LAUNCH;                                                                              \
        hipLaunchKernelGGL((block_attn_bwd_bf16_kernel<K, FULL>), grid, dim3(64 * K), lds, st, qhat, kvhat,      \
                           qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);                                 \
        break;                                                                                                   \
    }
    switch (nkt) {
        HEPT_BWDH_CASE(1)
        HEPT_BWDH_CASE(2)
        HEPT_BWDH_CASE(3)
        HEPT_BWDH_CASE(4)
        HEPT_BWDH_CASE(5)
        HEPT_BWDH_CASE(6)
        HEPT_BWDH_CASE(7)
        HEPT_BWDH_CASE(8)
        default:
            return HEPT_ERR_SHAPE;
    }
#undef HEPT_BWDH_CASE
    return hept_launch_status();
}

int block_attn_bwd_impl(bool split, const float* qhat, const float* kvhat, const int32_t* qpos, const int32_t* kpos,
                        const float* gacc, int N, int H, int D, int Tl, int B, float* dq_part, float* dkv_part,
                        void* stream) {
    if (!qhat || !kvhat || !qpos || !kpos || !gacc || !dq_part || !dkv_part) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || B < 1 || B > HEPT_MAX_BLOCK || N % B != 0 || D < 1 || D > 28)
        return HEPT_ERR_SHAPE;
    const int nb = N / B, nkt = (B + 31) / 32;
    const dim3 grid((unsigned)((size_t)Tl * nb * H));
    hipStream_t st = (hipStream_t)stream;
    const bool full = B == 32 * nkt;
    if (split)
        return full ? launch_bwd_split<true>(nkt, grid, st, qhat, kvhat, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb)
                    : launch_bwd_split<false>(nkt, grid, st, qhat, kvhat, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);
    return full ? launch_bwd<true>(nkt, grid, st, qhat, kvhat, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb)
                : launch_bwd<false>(nkt, grid, st, qhat, kvhat, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);
}

}  // namespace

extern "C" int hept_block_attn_bwd(const float* qhat, const float* kvhat, const int32_t* qpos, const int32_t* kpos,
                                   const float* gacc, int N, int H, int D, int Tl, int B, float* dq_part,
                                   float* dkv_part, void* stream) {
    return block_attn_bwd_impl(true, qhat, kvhat, qpos, kpos, gacc, N, H, D, Tl, B, dq_part, dkv_part, stream);
}

extern "C" int hept_block_attn_bwd_f32mfma(const float* qhat, const float* kvhat, const int32_t* qpos,
                                           const int32_t* kpos, const float* gacc, int N, int H, int D, int Tl, int B,
                                           float* dq_part, float* dkv_part, void* stream) {
    return block_attn_bwd_impl(false, qhat, kvhat, qpos, kpos, gacc, N, H, D, Tl, B, dq_part, dkv_part, stream);
}

extern "C" int hept_block_attn_bwd_bf16(const void* qhat, const void* kvhat, const int32_t* qpos, const int32_t* kpos,
                                        const float* gacc, int N, int H, int D, int Tl, int B, void* dq_part16,
                                        void* dkv_part16, void* stream) {
    unsigned int* dq_part = reinterpret_cast<unsigned int*>(dq_part16);
    unsigned int* dkv_part = reinterpret_cast<unsigned int*>(dkv_part16);
    if (!qhat || !kvhat || !qpos || !kpos || !gacc || !dq_part || !dkv_part) return HEPT_ERR_ARG;
    if (N < 1 || H < 1 || Tl < 1 || B < 1 || B > HEPT_MAX_BLOCK || N % B != 0 || D < 1 || D > 28)
        return HEPT_ERR_SHAPE;
    const int nb = N / B, nkt = (B + 31) / 32;
    const dim3 grid((unsigned)((size_t)Tl * nb * H));
    hipStream_t st = (hipStream_t)stream;
    const char* q = reinterpret_cast<const char*>(qhat);
    const char* kv = reinterpret_cast<const char*>(kvhat);
    return B == 32 * nkt ? launch_bwd_bf16<true>(nkt, grid, st, q, kv, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb)
                         : launch_bwd_bf16<false>(nkt, grid, st, q, kv, qpos, kpos, gacc, dq_part, dkv_part, N, H, D, B, nb);
}

namespace {
int bwd_reduce_impl(bool rows16, const void* dq_part, const void* dkv_part, int Tl, int N, int H, int D, int C,
                    const float* coords, int raw_size, float* dq, float* dk, float* dv, float* dcs, float* d_sqrt_w,
                    void* stream) {
    if (!dq_part || !dkv_part || !dq || !dk || !dv || !dcs) return HEPT_ERR_ARG;
    if (d_sqrt_w && !coords) return HEPT_ERR_ARG;
    if (Tl < 1 || N < 1 || H < 1 || D < 1 || C < 1 || D + C > 30 || raw_size < 0 || raw_size > N) return HEPT_ERR_SHAPE;
    if (d_sqrt_w && H * C > 64) return HEPT_ERR_SHAPE;
    // d_sqrt_w: the per-workgroup partial sums reuse dq_part as scratch (ceil(N/256) * 64 floats): refuse a cloud so
    // small that they would overrun its Tl * N * H * 32 elements (2 bytes each with bf16 gradient rows)
    if (d_sqrt_w && (size_t)((N + DSW_POINTS - 1) / DSW_POINTS) * 64 * sizeof(float) >
                        (size_t)Tl * N * H * 32 * (rows16 ? 2 : 4))
        return HEPT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)N * H * 32;
    if (rows16)
        hipLaunchKernelGGL(bwd_reduce16_kernel, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const unsigned int*>(dq_part),
                           reinterpret_cast<const unsigned int*>(dkv_part), Tl, N, H, D, C, raw_size, dq, dk, dv, dcs);
    else
        hipLaunchKernelGGL(bwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(dq_part), reinterpret_cast<const float*>(dkv_part), Tl, N, H, D,
                           C, raw_size, dq, dk, dv, dcs);
    if (d_sqrt_w) {
        // per-workgroup partial sums live in dq_part: the reduction above was its last reader ((N+255)/256 * 64 floats
        // of the Tl*N*H*32 values (f32 or bf16) it holds)
        float* partial = reinterpret_cast<float*>(const_cast<void*>(dq_part));
        const int n_wgs = (N + DSW_POINTS - 1) / DSW_POINTS;
        hipLaunchKernelGGL(dsw_kernel, dim3((unsigned)n_wgs), dim3(256), 0, st, dcs, coords, N, H, C, partial);
        hipLaunchKernelGGL(dsw_sum_kernel, dim3(64 / HEPT_FSUM_OUT), dim3(256), 0, st, partial, n_wgs, H * C, d_sqrt_w);
    }
    return hept_launch_status();
}
}  // namespace

extern "C" int hept_bwd_reduce(const float* dq_part, const float* dkv_part, int Tl, int N, int H, int D, int C,
                               const float* coords, int raw_size, float* dq, float* dk, float* dv, float* dcs,
                               float* d_sqrt_w, void* stream) {
    return bwd_reduce_impl(false, dq_part, dkv_part, Tl, N, H, D, C, coords, raw_size, dq, dk, dv, dcs, d_sqrt_w, stream);
}

extern "C" int hept_bwd_reduce16(const void* dq_part16, const void* dkv_part16, int Tl, int N, int H, int D, int C,
                                 const float* coords, int raw_size, float* dq, float* dk, float* dv, float* dcs,
                                 float* d_sqrt_w, void* stream) {
    return bwd_reduce_impl(true, dq_part16, dkv_part16, Tl, N, H, D, C, coords, raw_size, dq, dk, dv, dcs, d_sqrt_w, stream);
}
