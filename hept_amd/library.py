"""``torch.library`` registration of the operator ("torch.compile-safe custom op").

The reference model is run under ``torch.compile(model)`` in its example notebook
(``example/example.ipynb:161``).  Dynamo cannot trace through ``ctypes`` calls into
``libhept_hip.so``; registered as opaque custom ops with shape-only "fake" kernels the whole
operator becomes ONE node of the captured graph instead of a graph break.  ``HEPTAttention.forward``
routes through these ops only while a compiler is tracing; eager calls keep the direct path.
There is no CPU kernel behind any of them.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

__all__ = ["forward_op", "forward_src_op", "attn_block_op"]


@torch.library.custom_op("hept_amd::forward", mutates_args=(), device_types="cuda")
def forward_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, coords: torch.Tensor, codes: torch.Tensor,
               w_rpe_weight: torch.Tensor, alpha: torch.Tensor, out_weight: torch.Tensor,
               out_bias: Optional[torch.Tensor], block_size: int, w_per_dist: int, precision: str) -> torch.Tensor:
    return ops.forward(q, k, v, coords, codes, w_rpe_weight, alpha, out_weight, out_bias, block_size=block_size,
                       w_per_dist=w_per_dist, precision=precision)


@forward_op.register_fake
def _(q, k, v, coords, codes, w_rpe_weight, alpha, out_weight, out_bias, block_size, w_per_dist, precision):
    return q.new_empty((q.shape[0], out_weight.shape[0]), dtype=torch.float32)


@torch.library.custom_op("hept_amd::forward_src", mutates_args=(), device_types="cuda")
def forward_src_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, coords: torch.Tensor, eta_idx: torch.Tensor,
                   phi_idx: torch.Tensor, regions_h: torch.Tensor, raw_size: int, w_rpe_weight: torch.Tensor,
                   alpha: torch.Tensor, out_weight: torch.Tensor, out_bias: Optional[torch.Tensor], block_size: int,
                   w_per_dist: int, precision: str) -> torch.Tensor:
    return ops.forward_src(q, k, v, coords, (eta_idx, phi_idx), regions_h, raw_size, w_rpe_weight, alpha, out_weight,
                           out_bias, block_size=block_size, w_per_dist=w_per_dist, precision=precision)


@forward_src_op.register_fake
def _(q, k, v, coords, eta_idx, phi_idx, regions_h, raw_size, w_rpe_weight, alpha, out_weight, out_bias, block_size,
      w_per_dist, precision):
    return q.new_empty((q.shape[0], out_weight.shape[0]), dtype=torch.float32)


_ATTN_KEYS = ("norm1.weight", "norm1.bias", "w_q.weight", "w_k.weight", "w_v.weight", "w_rpe.weight",
              "attn.e2lsh.alpha", "attn.out_linear.weight", "attn.out_linear.bias", "norm2.weight", "norm2.bias",
              "ff.0.weight", "ff.0.bias", "ff.2.weight", "ff.2.bias")


@torch.library.custom_op("hept_amd::attn_block", mutates_args=(), device_types="cuda")
def attn_block_op(x: torch.Tensor, coords: torch.Tensor, codes: torch.Tensor, norm1_w: torch.Tensor,
                  norm1_b: torch.Tensor, w_q: torch.Tensor, w_k: torch.Tensor, w_v: torch.Tensor, w_rpe: torch.Tensor,
                  alpha: torch.Tensor, out_w: torch.Tensor, out_b: torch.Tensor, norm2_w: torch.Tensor,
                  norm2_b: torch.Tensor, ff1_w: torch.Tensor, ff1_b: torch.Tensor, ff2_w: torch.Tensor,
                  ff2_b: torch.Tensor, num_heads: int, block_size: int, w_per_dist: int, eps1: float, eps2: float,
                  precision: str) -> torch.Tensor:
    """The whole Attn block (reference ``example/transformer.py:154-165``, eval mode) as one graph node."""
    params = dict(zip(_ATTN_KEYS, (norm1_w, norm1_b, w_q, w_k, w_v, w_rpe, alpha, out_w, out_b, norm2_w, norm2_b,
                                   ff1_w, ff1_b, ff2_w, ff2_b)))
    return ops.attn_block_forward(x, coords, codes, params, num_heads=num_heads, block_size=block_size,
                                  w_per_dist=w_per_dist, eps1=eps1, eps2=eps2, precision=precision)


@attn_block_op.register_fake
def _(x, coords, codes, norm1_w, norm1_b, w_q, w_k, w_v, w_rpe, alpha, out_w, out_b, norm2_w, norm2_b, ff1_w, ff1_b,
      ff2_w, ff2_b, num_heads, block_size, w_per_dist, eps1, eps2, precision):
    return x.new_empty(x.shape, dtype=torch.float32)
