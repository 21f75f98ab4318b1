"""ctypes binding of the C ABI declared in ``include/hept_hip.h``.

There is deliberately no fallback: if the HIP library is missing or a symbol is
absent this raises, so a GPU box can never silently run a different code path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

# (kept in step with hept_amd/build.py, which is not imported here so that `python -m hept_amd.build` runs clean)
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libhept_hip.so")

ABI_VERSION = 20
PREC_F32, PREC_BF16, PREC_MIXED16, PREC_F32_MFMA, PREC_F32_DIFF = 0, 1, 2, 3, 4
ROW = 32
MAX_TABLES = 8
MAX_BLOCK = 256
PREP_GRID = 1024
TRANSPORT_RCCL, TRANSPORT_ONE_SIDED = 0, 1

_ERRORS = {
    1: "HEPT_ERR_SHAPE: unsupported or inconsistent sizes",
    2: "HEPT_ERR_LAUNCH: HIP reported a launch error",
    3: "HEPT_ERR_ARG: null pointer or workspace too small",
    4: "HEPT_ERR_COMM: RCCL unavailable or reported an error",
}

_P = c_void_p
# name -> (restype, argtypes); must list every symbol of include/hept_hip.h
SIGNATURES = {
    "hept_abi_version": (c_int, []),
    "hept_check_shape": (c_int, [c_int] * 6),
    "hept_workspace_bytes": (c_size_t, [c_int] * 7),
    "hept_rpe_scale": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "hept_prep_hash": (c_int, [_P] * 7 + [c_int] * 9 + [_P] * 6),
    "hept_sort_workspace_bytes": (c_size_t, [c_int] * 3),
    "hept_sort_tables": (c_int, [_P] * 4 + [c_int] * 5 + [_P] * 4),
    "hept_sort_tables_src": (c_int, [_P] * 6 + [c_int] * 5 + [_P] * 4),
    "hept_argsort_workspace_bytes": (c_size_t, [c_int, c_int]),
    "hept_segmented_argsort_ragged": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "hept_segmented_argsort": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "hept_block_attn": (c_int, [_P] * 4 + [c_int] * 6 + [_P, _P]),
    "hept_block_attn_heads": (c_int, [_P] * 4 + [c_int] * 11 + [_P, _P]),
    "hept_part_precision": (c_int, [c_int, c_int]),
    "hept_reduce_heads": (c_int, [_P] + [c_int] * 8 + [_P, c_int, _P]),
    "hept_combine_groups": (c_int, [_P] + [c_int] * 8 + [c_size_t] + [_P] * 4),
    "hept_partial_begin": (c_int, [_P] * 7 + [c_int] * 10 + [_P, c_size_t, _P]),
    "hept_partial_begin_src": (c_int, [_P] * 7 + [c_int] + [_P] * 2 + [c_int] * 10 + [_P, c_size_t, _P]),
    "hept_partial_heads": (c_int, [_P, c_size_t] + [c_int] * 11 + [_P, _P]),
    "hept_reduce_tables": (c_int, [_P] + [c_int] * 5 + [_P, c_int, _P]),
    "hept_combine_out": (c_int, [_P] + [c_int] * 7 + [_P] * 4),
    "hept_forward": (c_int, [_P] * 9 + [c_int] * 8 + [_P, c_size_t, _P, _P]),
    "hept_forward_partial": (c_int, [_P] * 7 + [c_int] * 11 + [_P, c_size_t, _P, _P]),
    "hept_prep_hash_fused": (c_int, [_P] * 3 + [c_float] + [_P] * 7 + [c_int] * 9 + [_P] * 6),
    "hept_combine_ffn": (c_int, [_P] + [c_int] * 7 + [_P] * 5 + [c_float] + [_P] * 6),
    "hept_attn_block_forward": (c_int, [_P] * 4 + [c_int] * 8 + [_P, c_size_t, _P, _P]),
    "hept_combine_bwd_scratch_bytes": (c_size_t, [c_int]),
    "hept_combine_bwd_scratch_bytes_shape": (c_size_t, [c_int] * 3),
    "hept_combine_bwd": (c_int, [_P] * 3 + [c_int] * 3 + [_P] * 4 + [c_size_t, _P]),
    "hept_forward_src": (c_int, [_P] * 7 + [c_int] + [_P] * 4 + [c_int] * 8 + [_P, c_size_t, _P, _P]),
    "hept_forward_partial_src": (c_int, [_P] * 7 + [c_int] + [_P] * 2 + [c_int] * 11 + [_P, c_size_t, _P, _P]),
    "hept_block_attn_bwd": (c_int, [_P] * 5 + [c_int] * 5 + [_P] * 3),
    "hept_block_attn_bwd_f32mfma": (c_int, [_P] * 5 + [c_int] * 5 + [_P] * 3),
    "hept_block_attn_bwd_bf16": (c_int, [_P] * 5 + [c_int] * 5 + [_P] * 3),
    "hept_bwd_reduce": (c_int, [_P, _P] + [c_int] * 5 + [_P, c_int] + [_P] * 6),
    "hept_bwd_reduce16": (c_int, [_P, _P] + [c_int] * 5 + [_P, c_int] + [_P] * 6),
    "hept_rpe_scale_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "hept_prepare_workspace_bytes": (c_size_t, [c_int] * 5),
    "hept_prepare_input": (c_int, [_P, c_int, _P, _P] + [c_int] * 4 + [_P] + [c_int] * 3 + [_P, c_size_t] + [_P] * 5),
    "hept_prepare_probe": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P]),
    "hept_comm_unique_id": (c_int, [_P]),
    "hept_comm_create": (c_int, [_P, c_int, c_int, _P]),
    "hept_comm_destroy": (c_int, [_P]),
    "hept_comm_rank": (c_int, [_P]),
    "hept_comm_world": (c_int, [_P]),
    "hept_comm_last_error": (ctypes.c_char_p, []),
    "hept_exchange_bytes": (c_size_t, [c_int] * 5),
    "hept_forward_sharded": (c_int, [_P] * 10 + [c_int] * 12 + [_P, c_size_t, _P, c_size_t, _P, _P]),
    "hept_forward_sharded_src": (c_int, [_P] * 8 + [c_int] + [_P] * 4 + [c_int] * 12 + [_P, c_size_t, _P, c_size_t, _P, _P]),
    "hept_comm_has_rccl": (c_int, [_P]),
    "hept_comm_create_local": (c_int, [c_int, c_int, _P]),
    "hept_p2p_bytes": (c_size_t, [c_int] * 5),
    "hept_comm_p2p_alloc": (c_int, [_P, c_size_t, _P]),
    "hept_comm_p2p_open": (c_int, [_P, _P]),
    "hept_comm_p2p_ready": (c_int, [_P, c_size_t]),
    "hept_comm_status": (c_int, [_P, _P]),
    "hept_comm_p2p_flags": (c_int, [_P, _P, _P]),
    "hept_comm_reset_status": (c_int, [_P]),
    "hept_comm_set_out_view": (c_int, [_P, c_int]),
    "hept_comm_out_view": (c_int, [_P, _P]),
    "hept_prepare_src_workspace_bytes": (c_size_t, [c_int]),
    "hept_prepare_input_src": (c_int, [_P, c_int, _P] + [c_int] * 3 + [_P, c_int, c_int, _P, c_size_t] + [_P] * 5),
    "hept_rows_wgrad_scratch_bytes": (c_size_t, [c_int, c_int]),
    "hept_rows_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "hept_ln_scratch_bytes": (c_size_t, [c_int]),
    "hept_ln_bwd": (c_int, [_P] * 4 + [c_float, c_int, c_int] + [_P] * 5 + [c_size_t, _P]),
    "hept_ln_ffn_fwd": (c_int, [_P] * 3 + [c_float] + [_P] * 4 + [c_int, c_int, _P, _P]),
    "hept_ln_ffn_bwd_scratch_bytes": (c_size_t, [c_int]),
    "hept_ln_ffn_bwd": (c_int, [_P] * 4 + [c_float] + [_P] * 4 + [c_int, c_int] + [_P] * 8 + [c_size_t, _P]),
    "hept_profile_enable": (c_int, [c_int, c_int]),
    "hept_profile_read": (c_int, [_P, _P]),
    "hept_profile_stride": (c_int, [c_int]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load ``libhept_hip.so`` (built in-tree by ``hept_amd.build``) and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m hept_amd.build` "
            "(or __graft_entry__.build()). hept_amd has no CPU/PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.hept_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libhept_hip.so ABI version {got}, Python binding expects {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


class AttnParams(ctypes.Structure):
    """``hept_attn_params`` of include/hept_hip.h."""

    _fields_ = [(name, c_void_p) for name in (
        "norm1_w", "norm1_b", "w_q", "w_k", "w_v", "w_rpe", "alpha", "out_w", "out_b", "norm2_w", "norm2_b",
        "ff1_w", "ff1_b", "ff2_w", "ff2_b")] + [("eps1", c_float), ("eps2", c_float)]


def check(rc: int, what: str) -> None:
    if rc != 0:
        detail = ""
        if rc == 4 and _lib is not None:
            detail = " (" + (_lib.hept_comm_last_error() or b"").decode(errors="replace") + ")"
        raise RuntimeError(f"{what} failed: {_ERRORS.get(rc, f'error code {rc}')}{detail}")
