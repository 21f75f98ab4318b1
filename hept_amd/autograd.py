"""Training path: autograd around the HIP kernels.

The reference trains through ``HEPTAttention.forward`` with plain autograd (``example/trainer.py:11-22``):
gradients reach ``query, key, value``, ``w_rpe.weight`` and ``out_linear``; hashing and sorting carry none
(``lsh_mapping`` is ``@torch.no_grad``, ``argsort`` yields integers; ``e2lsh.alpha`` is frozen).  Here the
non-trivial part — everything from the augmented rows to the table-summed partial rows — is one
``torch.autograd.Function`` whose forward and backward are HIP kernels (f32 tiles); the tiny
differentiable parameter math around it (``sqrt_w`` from ``w_rpe.weight``, the final divide and
``out_linear``) stays in torch so that its gradients come from autograd itself.
"""
from __future__ import annotations

import torch

from . import ops

__all__ = ["HeptPartialSums", "HeptPartialSumsFused", "HeptCombine", "LnFfn", "RpeScale", "rpe_scale_torch", "ReplicatedGrad",
           "sum_over_ranks"]


class ReplicatedGrad(torch.autograd.Function):
    """Identity on a tensor that every rank holds a copy of; the backward sums the ranks' gradient contributions
    (each rank differentiates only its own hash tables)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist

        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


class _SumOverRanks(torch.autograd.Function):
    """All-reduce (sum) whose result feeds the same computation on every rank: the gradient of the sum with respect to
    this rank's term is the (replicated) upstream gradient itself."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist

        y = x.contiguous().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


def sum_over_ranks(x, group):
    return _SumOverRanks.apply(x, group)


def rpe_scale_torch(w_rpe_weight: torch.Tensor, n_heads: int, head_dim: int, w_per_dist: int) -> torch.Tensor:
    """Differentiable ``sqrt(2 * sum_k exp(min(sum_d w, 50)))`` with column 0 duplicated; (H, C).
    Reference ``example/hept.py:48-54`` (view) and ``:22-23,25``."""
    w4 = w_rpe_weight.reshape(n_heads, head_dim, -1, w_per_dist)
    qw = w4.sum(dim=1).clamp(max=50).exp().sum(dim=-1)
    return torch.sqrt(2 * torch.cat([qw[:, :1], qw], dim=-1))


class RpeScale(torch.autograd.Function):
    """``w_rpe.weight`` -> sqrt_w (H, C): the HIP kernels ``hept_rpe_scale`` / ``hept_rpe_scale_bwd`` in place of the
    seven (forward) + eight (backward) tiny torch kernels of ``rpe_scale_torch`` (5 us of launch each)."""

    @staticmethod
    def forward(ctx, w_rpe_weight, n_heads, head_dim, w_per_dist):
        ctx.save_for_backward(w_rpe_weight)
        ctx.dims = (n_heads, head_dim, w_per_dist)
        return ops.rpe_scale(w_rpe_weight, n_heads, head_dim, w_per_dist)

    @staticmethod
    def backward(ctx, d_sqrt_w):
        (w,) = ctx.saved_tensors
        h, d, k = ctx.dims
        return ops.rpe_scale_bwd(w, d_sqrt_w.contiguous(), h, d, k).to(w.dtype), None, None, None


class HeptPartialSums(torch.autograd.Function):
    """(q, k, v, coords, sqrt_w) -> acc (N, H, 32) = sum over tables of [numer | denom | 0].

    ``tiles``: "fp32" (the reference's arithmetic; split-bf16 products in both directions) or "bf16" (the rows and
    kernels of the bf16 forward, one bf16 MFMA per product in the backward too -- ``HEPTAttention.train_tiles``).

    ``geo`` = (eta, phi, cfac, raw_size) selects the reference's src variant (``codes`` is then None): rows at and
    after ``raw_size`` are zero-filled in place by the reference (``src/models/attention/hept.py:89-91``), so no
    gradient flows into them.
    """

    @staticmethod
    def forward(ctx, q, k, v, coords, sqrt_w, alpha, codes, block_size, geo=None, f32_mfma=False, tiles="fp32"):
        n, hd = q.shape
        h = alpha.shape[0]
        d = hd // h
        # hashes and the sort walk the tables in chunks of HEPT_MAX_TABLES, as the inference entry points do (the
        # reference takes any n_hashes, example/hept.py:37-41); the rows are rewritten identically by every chunk
        from ._lib import MAX_TABLES

        n_tables = alpha.shape[2]
        qs, ks = [], []
        rows = None   # the row buffers of the first chunk: later chunks rewrite the same rows into them
        for c0 in range(0, n_tables, MAX_TABLES):
            tc = min(MAX_TABLES, n_tables - c0)
            if geo is None:
                ph = ops.prep_hash(q, k, v, coords, sqrt_w, alpha, codes, tiles, t0=c0, tl=tc, rows=rows)
                qp, kp = ops.sort_tables(ph["qproj"], ph["kproj"], codes, ph["minmax"], t0=c0)
            else:
                eta, phi, cfac, raw_size = geo
                ph = ops.prep_hash(q, k, v, coords, sqrt_w, alpha, None, tiles, t0=c0, tl=tc, raw_size=raw_size, rows=rows)
                qp, kp = ops.sort_tables_src(ph["qproj"], ph["kproj"], eta, phi, cfac, ph["minmax"], t0=c0)
            rows = (ph["qhat"], ph["kvhat"])
            qs.append(qp)
            ks.append(kp)
        qpos, kpos = (qs[0], ks[0]) if len(qs) == 1 else (torch.cat(qs), torch.cat(ks))
        part = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, d, block_size, f32_mfma=f32_mfma)
        acc = ops.reduce_tables(part, d)
        ctx.f32_mfma = f32_mfma
        ctx.save_for_backward(ph["qhat"], ph["kvhat"], qpos, kpos, coords, sqrt_w)
        ctx.dims = (d, coords.shape[1], block_size)
        ctx.raw_size = n if geo is None else int(geo[3])
        return acc

    @staticmethod
    def backward(ctx, gacc):
        qhat, kvhat, qpos, kpos, coords, sqrt_w = ctx.saved_tensors
        d, c, block_size = ctx.dims
        # scaled coordinates s[n,h,c] = sqrt_w[h,c] * coords[n,c]: d sqrt_w = sum_n dcs * coords comes out of the
        # reduction kernel (as a torch einsum it was a 48 x N GEMM that rocBLAS ran in 320 us, as a product + column
        # sum two kernels of 27 us); rows at and after raw_size (src variant padding) get zero gradients there too
        h = qhat.shape[0]
        if h * c <= 64:
            dq, dk, dv, dcs, dsw = ops.block_attn_bwd(qhat, kvhat, qpos, kpos, gacc.contiguous(), d, c, block_size,
                                                      f32_mfma=ctx.f32_mfma, coords=coords, raw_size=ctx.raw_size)
        else:  # more (head, coordinate) columns than the reduction kernel's 64 lanes: the column sum in torch
            dq, dk, dv, dcs = ops.block_attn_bwd(qhat, kvhat, qpos, kpos, gacc.contiguous(), d, c, block_size,
                                                 f32_mfma=ctx.f32_mfma, raw_size=ctx.raw_size)
            dsw = (dcs * coords[:, None, :]).sum(dim=0)
        if not ctx.needs_input_grad[4]:
            dsw = None
        dcoords = (dcs * sqrt_w[None]).sum(dim=1) if ctx.needs_input_grad[3] else None
        return dq, dk, dv, dcoords, dsw, None, None, None, None, None, None


class HeptPartialSumsFused(torch.autograd.Function):
    """The ``Attn`` block's front end and the operator as ONE autograd node (training mode, SURVEY.md §8 f-4 x f-2):
    (x, norm1, w_q / w_k / w_v, coords, sqrt_w) -> acc (N, H, 32).

    Forward: ``hept_prep_hash_fused`` -- LayerNorm and the three bias-free projections are computed while the rows of
    the operator are staged (reference ``example/transformer.py:155-156``); q, k, v (3 x 46 MB at tracking-60k) are
    never written or read back -- then the sort, the block attention and the table sum as in :class:`HeptPartialSums`.
    Backward: the HIP backward of the block attention yields dq, dk, dv; the small dense part behind them (three
    (N, 192) x (192, 24) products for d LayerNorm(x), three transposed ones for the weight gradients, the LayerNorm
    backward) is plain torch on the saved ``x`` -- the normalised rows are recomputed, nothing of size (N, 192) is kept
    between the passes."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w_q, w_k, w_v, coords, sqrt_w, alpha, codes, block_size, f32_mfma=False,
                tiles="fp32"):
        from ._lib import MAX_TABLES

        n, d = x.shape
        n_tables = alpha.shape[2]
        qs, ks = [], []
        rows = None   # see HeptPartialSums.forward
        for c0 in range(0, n_tables, MAX_TABLES):
            tc = min(MAX_TABLES, n_tables - c0)
            ph = ops.prep_hash_fused(x, ln_w, ln_b, eps, w_q, w_k, w_v, coords, sqrt_w, alpha, codes, tiles, t0=c0, tl=tc,
                                     rows=rows)
            rows = (ph["qhat"], ph["kvhat"])
            qp, kp = ops.sort_tables(ph["qproj"], ph["kproj"], codes, ph["minmax"], t0=c0)
            qs.append(qp)
            ks.append(kp)
        qpos, kpos = (qs[0], ks[0]) if len(qs) == 1 else (torch.cat(qs), torch.cat(ks))
        part = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, d, block_size, f32_mfma=f32_mfma)
        acc = ops.reduce_tables(part, d)
        ctx.f32_mfma = f32_mfma
        ctx.save_for_backward(ph["qhat"], ph["kvhat"], qpos, kpos, coords, sqrt_w, x, ln_w, ln_b, w_q, w_k, w_v)
        ctx.dims = (d, coords.shape[1], block_size, float(eps))
        return acc

    @staticmethod
    def backward(ctx, gacc):
        qhat, kvhat, qpos, kpos, coords, sqrt_w, x, ln_w, ln_b, w_q, w_k, w_v = ctx.saved_tensors
        d, c, block_size, eps = ctx.dims
        dq, dk, dv, dcs, dsw = ops.block_attn_bwd(qhat, kvhat, qpos, kpos, gacc.contiguous(), d, c, block_size,
                                                  f32_mfma=ctx.f32_mfma, coords=coords, raw_size=x.shape[0])
        need = ctx.needs_input_grad
        dx = dlw = dlb = dwq = dwk = dwv = None
        if any(need[i] for i in (0, 1, 2, 4, 5, 6)):   # (a frozen front end: nothing behind dq, dk, dv is computed)
            dxn = dq @ w_q + dk @ w_k + dv @ w_v                   # d of the three Linear(D, H*D): (N,192) x (192,24)
            # LayerNorm backward + the normalised rows in one kernel; the three weight gradients dq^T.xn ... are 192 x 24
            # outputs reduced over all points -- ~140 us each in rocBLAS at 60k points, ~15 us here (csrc/block_train.hip)
            dx, xn_d, dlw, dlb = ops.ln_bwd(x, dxn, ln_w, ln_b, eps)                   # example/transformer.py:155
            dwq = ops.rows_wgrad(dq, xn_d) if need[4] else None
            dwk = ops.rows_wgrad(dk, xn_d) if need[5] else None
            dwv = ops.rows_wgrad(dv, xn_d) if need[6] else None
        dcoords = (dcs * sqrt_w[None]).sum(dim=1) if need[7] else None
        return (dx if need[0] else None, dlw if need[1] else None, dlb if need[2] else None, None,
                dwq if need[4] else None, dwk if need[5] else None, dwv if need[6] else None, dcoords,
                dsw if need[8] else None, None, None, None, None, None)


class HeptCombine(torch.autograd.Function):
    """acc (N, H, 32), out_linear.weight, out_linear.bias -> out (N, D): the cross-table divide and ``out_linear``
    (reference ``example/hept.py:79-80``) as HIP kernels in both directions (``combine_out`` / ``combine_bwd``)."""

    @staticmethod
    def forward(ctx, acc, weight, bias):
        d = weight.shape[0]
        out = ops.combine_out(acc, d, weight, bias)
        ctx.save_for_backward(acc, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        acc, weight = ctx.saved_tensors
        gacc, dw, db = ops.combine_bwd(acc, g_out.contiguous(), weight, need_bias=ctx.has_bias)
        return gacc, dw, db


class LnFfn(torch.autograd.Function):
    """``ff(norm2(x1))`` of the ``Attn`` block (reference ``example/transformer.py:162``) as one autograd node with HIP
    kernels in both directions (``hept_ln_ffn_fwd`` / ``hept_ln_ffn_bwd``): per point a 24-wide LayerNorm and two
    24 x 24 layers; the parameter gradients are fixed-order reductions over the points."""

    @staticmethod
    def forward(ctx, x1, ln_w, ln_b, eps, w1, b1, w2, b2):
        ctx.save_for_backward(x1, ln_w, ln_b, w1, b1, w2, b2)
        ctx.eps = float(eps)
        return ops.ln_ffn_fwd(x1, ln_w, ln_b, eps, w1, b1, w2, b2)

    @staticmethod
    def backward(ctx, d_out):
        x1, ln_w, ln_b, w1, b1, w2, b2 = ctx.saved_tensors
        dx1, dlw, dlb, dw1, db1, dw2, db2 = ops.ln_ffn_bwd(x1, d_out.contiguous(), ln_w, ln_b, ctx.eps, w1, b1, w2, b2)
        return dx1, dlw, dlb, None, dw1, db1, dw2, db2
