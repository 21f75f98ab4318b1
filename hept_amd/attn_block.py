"""``Attn``: the transformer block around the operator, a drop-in for the reference's
``Attn`` module (``example/transformer.py:131-165``).

Same constructor (``coords_dim`` positional, model-config kwargs), same ``forward(x, kwargs)`` with
``kwargs = {"coords", "combined_shifts"}`` from ``prepare_input``, same sub-module and state-dict names
(``w_q, w_k, w_v, attn.out_linear, attn.e2lsh, norm1, norm2, ff.0, ff.2, w_rpe``), so the ``attns.{i}.*`` entries
of a reference checkpoint load with ``strict=True``.

In eval mode under ``torch.no_grad()`` the whole block is ONE C call (``hept_attn_block_forward``): LayerNorm and the
q/k/v projections are computed while the rows of the operator are staged (q, k, v never exist in HBM: 138 MB written
and read back per layer at tracking-60k in the unfused form), and the residual, ``norm2`` and the feed-forward run in
the epilogue of the combine kernel.  In training mode (dropout active, gradients) the rest of the block is composed of
torch modules exactly like the reference, but ``norm1`` and the three projections are folded into the operator's row
builder as one autograd node (``autograd.HeptPartialSumsFused``): q, k, v are never written to HBM in the forward, and
the backward adds the small dense products behind the HIP gradients of the block attention.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .hept import HEPTAttention

__all__ = ["Attn"]


class Attn(nn.Module):
    def __init__(self, coords_dim, *, precision: str = "fp32", **kwargs):
        super().__init__()
        self.dim_per_head = kwargs["h_dim"]
        self.num_heads = kwargs["num_heads"]

        self.w_q = nn.Linear(self.dim_per_head, self.dim_per_head * self.num_heads, bias=False)
        self.w_k = nn.Linear(self.dim_per_head, self.dim_per_head * self.num_heads, bias=False)
        self.w_v = nn.Linear(self.dim_per_head, self.dim_per_head * self.num_heads, bias=False)

        self.attn = HEPTAttention(self.dim_per_head + coords_dim, precision=precision, **kwargs)

        self.dropout = nn.Dropout(0.1)
        self.norm1 = nn.LayerNorm(self.dim_per_head)
        self.norm2 = nn.LayerNorm(self.dim_per_head)
        self.ff = nn.Sequential(
            nn.Linear(self.dim_per_head, self.dim_per_head),
            nn.ReLU(),
            nn.Linear(self.dim_per_head, self.dim_per_head),
        )
        self.w_rpe = nn.Linear(kwargs["num_w_per_dist"] * (coords_dim - 1), self.num_heads * self.dim_per_head)
        self._workspace = None

    def _fused_ok(self, x) -> bool:
        return (x.is_cuda and not self.training and not torch.is_grad_enabled() and self.dim_per_head == 24
                and self.num_heads == 8 and self.attn.sharding is None)

    # training / grad-enabled calls: LayerNorm + projections + operator as one autograd node (False: compose modules)
    fuse_training = True

    def forward(self, x, kwargs):
        if not self._fused_ok(x):
            # reference composition, example/transformer.py:154-165 (training: dropout + autograd)
            if self.fuse_training and torch.is_grad_enabled() and self.attn._train_fused_ok(x, kwargs):
                # the same composition with norm1 / w_q / w_k / w_v folded into the operator's row builder: q, k, v
                # never exist in HBM, gradients reach the same parameters (autograd.HeptPartialSumsFused)
                aggr_out = self.attn._forward_train_fused(x, self.norm1, self.w_q, self.w_k, self.w_v,
                                                          w_rpe=self.w_rpe, **kwargs)
            else:
                x_normed = self.norm1(x)
                q, k, v = self.w_q(x_normed), self.w_k(x_normed), self.w_v(x_normed)
                aggr_out = self.attn(q, k, v, pe=kwargs["coords"], w_rpe=self.w_rpe, **kwargs)
            x = x + self.dropout(aggr_out)
            if (self.fuse_training and torch.is_grad_enabled() and x.is_cuda and self.dim_per_head == 24
                    and x.dtype == torch.float32):
                from .autograd import LnFfn   # norm2 + ff.0 + ReLU + ff.2 as one autograd node (HIP both ways)

                ff_output = LnFfn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, self.ff[0].weight,
                                        self.ff[0].bias, self.ff[2].weight, self.ff[2].bias)
            else:
                ff_output = self.ff(self.norm2(x))
            return x + self.dropout(ff_output)
        a = self.attn
        if torch.compiler.is_compiling():
            # one opaque graph node (hept_amd/library.py) instead of a ctypes call Dynamo cannot trace
            from .library import attn_block_op

            y = attn_block_op(x.float(), kwargs["coords"].float(), kwargs["combined_shifts"], self.norm1.weight,
                              self.norm1.bias, self.w_q.weight, self.w_k.weight, self.w_v.weight, self.w_rpe.weight,
                              a.e2lsh.alpha, a.out_linear.weight, a.out_linear.bias, self.norm2.weight,
                              self.norm2.bias, self.ff[0].weight, self.ff[0].bias, self.ff[2].weight,
                              self.ff[2].bias, self.num_heads, a.block_size, a.num_w_per_dist, self.norm1.eps,
                              self.norm2.eps, a.precision)
            return y.to(x.dtype)
        n = x.shape[0]
        c = kwargs["coords"].shape[1]
        need = ops.workspace_bytes(n, self.num_heads, self.dim_per_head, c, a.n_hashes, a.block_size, a.precision)
        ws = self._workspace
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = self._workspace = torch.empty(need, device=x.device, dtype=torch.uint8)
        params = {
            "norm1.weight": self.norm1.weight, "norm1.bias": self.norm1.bias, "w_q.weight": self.w_q.weight,
            "w_k.weight": self.w_k.weight, "w_v.weight": self.w_v.weight,
            # the weight itself: sqrt_w (H, C) is computed inside the row builder on every call (no cached copy that an
            # in-place update of the parameter could leave stale)
            "w_rpe.weight": self.w_rpe.weight,
            "attn.e2lsh.alpha": a.e2lsh.alpha, "attn.out_linear.weight": a.out_linear.weight,
            "attn.out_linear.bias": a.out_linear.bias, "norm2.weight": self.norm2.weight,
            "norm2.bias": self.norm2.bias, "ff.0.weight": self.ff[0].weight, "ff.0.bias": self.ff[0].bias,
            "ff.2.weight": self.ff[2].weight, "ff.2.bias": self.ff[2].bias,
        }
        y = ops.attn_block_forward(x.float(), kwargs["coords"].float(), kwargs["combined_shifts"], params,
                                   num_heads=self.num_heads, block_size=a.block_size, w_per_dist=a.num_w_per_dist,
                                   eps1=self.norm1.eps, eps2=self.norm2.eps, precision=a.precision, workspace=ws)
        return y.to(x.dtype)
