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

__all__ = ["forward_op", "forward_src_op"]


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
