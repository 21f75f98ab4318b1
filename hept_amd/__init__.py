"""hept_amd — MI355X (gfx950) implementation of HEPT's LSH block-attention hot path.

``HEPTAttention`` is a drop-in for the reference module (``example/hept.py``);
``prepare_input`` mirrors the caller-side preparation (``example/transformer.py``);
``Attn`` is the fused transformer block around the operator (``example/transformer.py:131-165``).
The compute path is the HIP library ``csrc/libhept_hip.so`` (C ABI in
``include/hept_hip.h``); there is no CPU or eager-PyTorch fallback.
"""
from .attn_block import Attn
from .hept import E2LSH, HEPTAttention
from .prep import (bit_shift, get_regions, pad_and_unpad, prepare_input, prepare_input_hip, prepare_input_src,
                   quantile_partition)

__all__ = [
    "HEPTAttention", "E2LSH", "Attn", "prepare_input", "prepare_input_hip", "prepare_input_src", "get_regions",
    "quantile_partition", "bit_shift", "pad_and_unpad",
]
