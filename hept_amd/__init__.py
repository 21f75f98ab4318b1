"""hept_amd — MI355X (gfx950) implementation of HEPT's LSH block-attention hot path.

``HEPTAttention`` is a drop-in for the reference module (``example/hept.py``);
``prepare_input`` mirrors the caller-side preparation (``example/transformer.py``).
The compute path is the HIP library ``csrc/libhept_hip.so`` (C ABI in
``include/hept_hip.h``); there is no CPU or eager-PyTorch fallback.
"""
from .hept import E2LSH, HEPTAttention
from .prep import bit_shift, get_regions, pad_and_unpad, prepare_input, prepare_input_hip, quantile_partition

__all__ = [
    "HEPTAttention", "E2LSH", "prepare_input", "prepare_input_hip", "get_regions", "quantile_partition", "bit_shift", "pad_and_unpad",
]
