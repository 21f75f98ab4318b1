"""Torch-tensor front end of the C ABI: argument checks, workspace from the caching allocator, current stream.

PyTorch is plumbing here (device memory + streams); all arithmetic runs in the
HIP kernels of ``csrc/``.  Stage-level functions exist so that the parity tests
can check every stage against the oracle and inject permutations.
"""
from __future__ import annotations

import functools
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import PREC_BF16, PREC_F32, PREC_F32_DIFF, PREC_F32_MFMA, PREC_MIXED16

__all__ = [
    "precision_code", "rpe_scale", "prep_hash", "sort_tables", "block_attn", "reduce_tables", "combine_out",
    "forward", "forward_partial", "workspace_bytes", "profile_enable", "profile_read", "unpack_part",
    "segmented_argsort", "block_attn_bwd", "sort_tables_src", "forward_src", "forward_partial_src", "geo_args",
    "packed_partials", "prep_hash_fused", "combine_ffn", "attn_block_forward", "combine_bwd", "rpe_scale_bwd",
    "partial_begin", "partial_heads", "combine_groups", "forward_sharded", "rows_wgrad", "ln_bwd", "ln_ffn_fwd",
    "ln_ffn_bwd",
]


def precision_code(precision) -> int:
    if isinstance(precision, int) and not isinstance(precision, bool) and precision in (PREC_F32, PREC_BF16, PREC_MIXED16, PREC_F32_MFMA, PREC_F32_DIFF):
        return precision
    if precision == "mixed16":
        return PREC_MIXED16
    if precision in ("fp32", "f32") or precision is torch.float32:
        return PREC_F32
    if precision == "fp32_mfma":
        return PREC_F32_MFMA
    if precision == "fp32_diff":
        return PREC_F32_DIFF
    if precision == "bf16" or precision is torch.bfloat16:
        return PREC_BF16
    raise ValueError(f"precision must be 'fp32', 'bf16', 'mixed16', 'fp32_mfma' or 'fp32_diff', got {precision!r}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr(device) -> int:
    """The current HIP stream of ``device`` as the integer the C ABI takes.  ``torch._C._cuda_getCurrentRawStream`` is one C
    call (0.2 us); ``torch.cuda.current_stream(...).cuda_stream`` builds a Stream object first (2 us of the ~30 us a forward
    costs the host) and is the fallback where the private entry point is missing."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t: torch.Tensor) -> int:
    return current_stream_ptr(t.device)


def _on_device(fn):
    """The C ABI launches on the thread's *current* HIP device: make that the device of the first GPU tensor argument
    for the duration of the call (a module on cuda:1 called while cuda:0 is current)."""

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = next((a.device for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)

    return wrapped


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_cuda and t.is_contiguous() and not (t.data_ptr() & 15):
        return t   # the usual case, checked first: this function runs a dozen times per forward
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (hept_amd has no CPU path); got device {t.device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    t = t.contiguous()
    # the kernels read rows and weight columns as 16-byte pieces (include/hept_hip.h: hept_combine_out refuses other
    # bases): a contiguous VIEW at an odd storage offset gets an aligned copy here instead of an error there
    return t.clone() if t.data_ptr() % 16 else t


def _dims(q: torch.Tensor, coords: torch.Tensor, alpha: torch.Tensor) -> Tuple[int, int, int, int, int]:
    n, hd = q.shape
    h, e, t = alpha.shape
    c = coords.shape[1]
    if hd % h != 0 or hd // h + c != e:
        raise ValueError(f"inconsistent sizes: q {tuple(q.shape)}, coords {tuple(coords.shape)}, alpha {tuple(alpha.shape)}")
    return n, h, hd // h, c, t


@functools.lru_cache(maxsize=256)
def _workspace_bytes_cached(n: int, h: int, d: int, c: int, tl: int, b: int, prec: int) -> int:
    # a pure function of the sizes (no tensor, no parameter): remembered per shape -- one ctypes round trip less per call
    return int(_lib.load().hept_workspace_bytes(n, h, d, c, tl, b, prec))


@functools.lru_cache(maxsize=256)
def _check_shape_cached(n: int, h: int, d: int, c: int, tl: int, b: int) -> int:
    return int(_lib.load().hept_check_shape(n, h, d, c, tl, b))


def workspace_bytes(n, h, d, c, tl, b, precision) -> int:
    return _workspace_bytes_cached(int(n), int(h), int(d), int(c), int(tl), int(b), precision_code(precision))


def exchange_bytes(n, h, d, world, precision) -> int:
    return int(_lib.load().hept_exchange_bytes(n, h, d, world, precision_code(precision)))


def p2p_bytes(n, h, d, world, precision) -> int:
    return int(_lib.load().hept_p2p_bytes(n, h, d, world, precision_code(precision)))


@_on_device
def rpe_scale(w_rpe_weight: torch.Tensor, n_heads: int, head_dim: int, w_per_dist: int) -> torch.Tensor:
    lib = _lib.load()
    w = _f32c(w_rpe_weight, "w_rpe.weight")
    c = w.shape[1] // w_per_dist + 1
    out = torch.empty(n_heads, c, device=w.device, dtype=torch.float32)
    _lib.check(lib.hept_rpe_scale(w.data_ptr(), n_heads, head_dim, c, w_per_dist, out.data_ptr(), _stream(w)),
               "hept_rpe_scale")
    return out


@_on_device
def rpe_scale_bwd(w_rpe_weight: torch.Tensor, d_sqrt_w: torch.Tensor, n_heads: int, head_dim: int,
                  w_per_dist: int) -> torch.Tensor:
    """Gradient of ``rpe_scale`` with respect to ``w_rpe.weight``: (H*D, (C-1)*K) from d_sqrt_w (H, C)."""
    lib = _lib.load()
    w = _f32c(w_rpe_weight, "w_rpe.weight")
    g = _f32c(d_sqrt_w, "d_sqrt_w")
    c = w.shape[1] // w_per_dist + 1
    out = torch.empty_like(w)
    _lib.check(lib.hept_rpe_scale_bwd(w.data_ptr(), g.data_ptr(), n_heads, head_dim, c, w_per_dist, out.data_ptr(),
                                      _stream(w)), "hept_rpe_scale_bwd")
    return out


@_on_device
def prep_hash(q, k, v, coords, sqrt_w, alpha, codes, precision="fp32", t0: int = 0, tl: Optional[int] = None,
              raw_size: Optional[int] = None, rows: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """``rows`` = (qhat, kvhat) of an earlier call with the same inputs and precision: the row buffers are reused
    (callers that walk more than HEPT_MAX_TABLES tables in chunks rebuild identical rows for every chunk)."""
    lib = _lib.load()
    q, k, v, coords, sqrt_w, alpha = (_f32c(x, nm) for x, nm in
                                      ((q, "query"), (k, "key"), (v, "value"), (coords, "coords"),
                                       (sqrt_w, "sqrt_w"), (alpha, "alpha")))
    n, h, d, c, t = _dims(q, coords, alpha)
    if tuple(sqrt_w.shape) != (h, c):   # the kernel indexes it as (H, C): a mismatch would be an out-of-bounds read
        raise ValueError(f"sqrt_w must have shape {(h, c)}, got {tuple(sqrt_w.shape)}")
    if codes is not None:
        if codes.dtype != torch.int64 or not codes.is_cuda or tuple(codes.shape) != (t, h, n):
            raise ValueError(f"combined_shifts must be an int64 GPU tensor of shape {(t, h, n)}")
        codes = codes.contiguous()
    raw_size = n if raw_size is None else int(raw_size)
    tl = t - t0 if tl is None else tl
    prec = precision_code(precision)
    # dtype tag of the row buffers: bf16 / f16 (mixed16: q^,k^ halves are fp16, the v half of kvhat is bf16) / f32
    tile = {PREC_F32: torch.float32, PREC_BF16: torch.bfloat16, PREC_MIXED16: torch.float16,
            PREC_F32_MFMA: torch.float32, PREC_F32_DIFF: torch.float32}[prec]
    dev = q.device
    if rows is not None:
        qhat, kvhat = rows
        if qhat.dtype != tile or tuple(qhat.shape) != (h, n, 32) or kvhat.dtype != tile or tuple(kvhat.shape) != (h, n, 64):
            raise ValueError("rows: buffers of another shape or precision")
    else:
        qhat = torch.empty(h, n, 32, device=dev, dtype=tile)
        kvhat = torch.empty(h, n, 64, device=dev, dtype=tile)
    qproj = torch.empty(tl, h, n, device=dev, dtype=torch.float32)
    kproj = torch.empty(tl, h, n, device=dev, dtype=torch.float32)
    minmax = torch.empty(tl, h, _lib.PREP_GRID, 4, device=dev, dtype=torch.float32)
    _lib.check(lib.hept_prep_hash(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), sqrt_w.data_ptr(),
                                  alpha.data_ptr(), codes.data_ptr() if codes is not None else None, n, raw_size, h, d, c,
                                  t, t0, tl, prec, qhat.data_ptr(), kvhat.data_ptr(),
                                  qproj.data_ptr(), kproj.data_ptr(), minmax.data_ptr(), _stream(q)),
               "hept_prep_hash")
    return dict(qhat=qhat, kvhat=kvhat, qproj=qproj, kproj=kproj, minmax=minmax)


@_on_device
def sort_tables(qproj, kproj, codes, minmax, t0: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stable ascending permutations (Tl,H,N) int32 of ``proj + float(code) * span`` for q and k."""
    lib = _lib.load()
    tl, h, n = qproj.shape
    if codes.dtype != torch.int64 or not codes.is_cuda:
        raise TypeError("combined_shifts must be an int64 GPU tensor")
    codes = codes.contiguous()
    t = codes.shape[0]
    ws = torch.empty(int(lib.hept_sort_workspace_bytes(n, h, tl)), device=qproj.device, dtype=torch.uint8)
    pos = torch.empty(2, tl, h, n, device=qproj.device, dtype=torch.int32)
    _lib.check(lib.hept_sort_tables(qproj.data_ptr(), kproj.data_ptr(), codes.data_ptr(), minmax.data_ptr(), n, h, t,
                                    t0, tl, ws.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), _stream(qproj)),
               "hept_sort_tables")
    return pos[0], pos[1]


@_on_device
def sort_tables_src(qproj, kproj, eta_idx, phi_idx, cfac, minmax, t0: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """src-variant keys (hash + get_geo_shift): stable ascending permutations (Tl,H,N) int32 for q and k."""
    lib = _lib.load()
    tl, h, n = qproj.shape
    eta_idx, phi_idx, cfac = (_f32c(x, nm) for x, nm in ((eta_idx, "region_indices[0]"), (phi_idx, "region_indices[1]"),
                                                          (cfac, "cfac")))
    t = eta_idx.numel() // (h * n)
    ws = torch.empty(int(lib.hept_sort_workspace_bytes(n, h, tl)), device=qproj.device, dtype=torch.uint8)
    pos = torch.empty(2, tl, h, n, device=qproj.device, dtype=torch.int32)
    _lib.check(lib.hept_sort_tables_src(qproj.data_ptr(), kproj.data_ptr(), eta_idx.data_ptr(), phi_idx.data_ptr(),
                                        cfac.data_ptr(), minmax.data_ptr(), n, h, t, t0, tl, ws.data_ptr(),
                                        pos[0].data_ptr(), pos[1].data_ptr(), _stream(qproj)), "hept_sort_tables_src")
    return pos[0], pos[1]


@_on_device
def segmented_argsort(keys: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stable ascending argsort of every row of a float32 (S, L) GPU tensor (+inf pads sort last); int32 (S, L).

    With ``lens`` (int32 (S,), on the GPU) only the first ``lens[s]`` keys of row s take part and only
    ``pos[s, :lens[s]]`` is defined (the ragged form ``prepare_input`` uses for clouds of unequal size)."""
    lib = _lib.load()
    keys = _f32c(keys, "keys")
    s_, l_ = keys.shape
    ws = torch.empty(int(lib.hept_argsort_workspace_bytes(s_, l_)), device=keys.device, dtype=torch.uint8)
    pos = torch.empty(s_, l_, device=keys.device, dtype=torch.int32)
    if lens is None:
        rc = lib.hept_segmented_argsort(keys.data_ptr(), s_, l_, ws.data_ptr(), pos.data_ptr(), _stream(keys))
    else:
        if lens.dtype != torch.int32 or lens.shape != (s_,) or lens.device != keys.device:
            raise ValueError("lens must be an int32 (S,) tensor on the keys' device")
        lens = lens.contiguous()
        rc = lib.hept_segmented_argsort_ragged(keys.data_ptr(), s_, l_, lens.data_ptr(), ws.data_ptr(), pos.data_ptr(),
                                               _stream(keys))
    _lib.check(rc, "hept_segmented_argsort")
    return pos


@_on_device
def block_attn(qhat, kvhat, qpos, kpos, head_dim: int, block_size: int, f32_mfma: bool = False) -> torch.Tensor:
    """Per-table partial rows (Tl, N, H, row).  fp32 tiles: row = 32 f32 (numer in [:D], denom (+1e-20) at [D]);
    bf16 tiles with D == 24: packed row = 16 int32 dwords (24 bf16 numer | f32 denom | 0); see ``unpack_part``.
    ``f32_mfma`` runs f32 tiles on the native f32 MFMA instead of the split-bf16 products; ``f32_mfma="diff"`` the
    difference form of the coordinate columns on top of that (``precision="fp32_diff"``)."""
    lib = _lib.load()
    h, n, _ = qhat.shape
    tl = qpos.shape[0]
    prec = {torch.float32: PREC_F32, torch.bfloat16: PREC_BF16, torch.float16: PREC_MIXED16}[qhat.dtype]
    if f32_mfma and prec == PREC_F32:
        prec = PREC_F32_DIFF if f32_mfma == "diff" else PREC_F32_MFMA
    packed = lib.hept_part_precision(prec, head_dim) == PREC_BF16
    qpos = qpos.to(torch.int32).contiguous()
    kpos = kpos.to(torch.int32).contiguous()
    part = torch.empty(tl, n, h, 16 if packed else 32, device=qhat.device, dtype=torch.int32 if packed else torch.float32)
    _lib.check(lib.hept_block_attn(qhat.data_ptr(), kvhat.data_ptr(), qpos.data_ptr(), kpos.data_ptr(), n, h,
                                   head_dim, tl, block_size, prec, part.data_ptr(), _stream(qhat)),
               "hept_block_attn")
    return part


def _part_prec(part: torch.Tensor) -> int:
    return PREC_BF16 if part.dtype == torch.int32 else PREC_F32


def unpack_part(part: torch.Tensor) -> torch.Tensor:
    """Widen packed partial rows (..., 16) int32 to the f32 row format (..., 32); f32 rows pass through."""
    if part.dtype != torch.int32:
        return part
    lo = (part[..., :12] << 16).view(torch.float32)
    hi = (part[..., :12] & -65536).view(torch.float32)
    out = torch.zeros(*part.shape[:-1], 32, device=part.device, dtype=torch.float32)
    out[..., 0:24:2] = lo
    out[..., 1:24:2] = hi
    out[..., 24] = part[..., 12].view(torch.float32)
    return out


@_on_device
def reduce_tables(part: torch.Tensor, head_dim: int = 24, packed: bool = False) -> torch.Tensor:
    """Sum of the per-table partial rows: f32 rows (N,H,32), or with ``packed`` (packed input only) packed rows
    (N,H,16) int32 again -- the form table sharding sends over xGMI."""
    lib = _lib.load()
    tl, n, h, _ = part.shape
    if packed and part.dtype != torch.int32:
        raise TypeError("packed sums need packed partial rows (16-bit tiles with head_dim 24)")
    acc = torch.empty(n, h, 16 if packed else 32, device=part.device, dtype=torch.int32 if packed else torch.float32)
    _lib.check(lib.hept_reduce_tables(part.data_ptr(), _part_prec(part), tl, n, h, head_dim, acc.data_ptr(),
                                      PREC_BF16 if packed else PREC_F32, _stream(part)), "hept_reduce_tables")
    return acc


@_on_device
def combine_out(part: torch.Tensor, head_dim: int, out_weight, out_bias, n0: int = 0, n_count: Optional[int] = None) -> torch.Tensor:
    """(sum_t numer / sum_t denom) -> Linear(H*D -> D) for points [n0, n0+n_count); part is (Tl,N,H,row) or (N,H,32)."""
    lib = _lib.load()
    if part.dim() == 3:
        part = part.unsqueeze(0)
    tl, n, h, _ = part.shape
    n_count = n - n0 if n_count is None else n_count
    w = _f32c(out_weight, "out_linear.weight")
    b = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    out = torch.empty(n_count, head_dim, device=part.device, dtype=torch.float32)
    _lib.check(lib.hept_combine_out(part.data_ptr(), _part_prec(part), tl, n, h, head_dim, n0, n_count, w.data_ptr(),
                                    b.data_ptr() if b is not None else None, out.data_ptr(), _stream(part)),
               "hept_combine_out")
    return out


@_on_device
def block_attn_bwd(qhat, kvhat, qpos, kpos, gacc, head_dim: int, coords_dim: int, block_size: int,
                   f32_mfma: bool = False, coords: Optional[torch.Tensor] = None, raw_size: Optional[int] = None):
    """Backward of block_attn + reduce_tables for f32 (or bf16) tiles: gradient rows gacc (N,H,32) -> dq, dk, dv (N, H*D)
    and dcs (N, H, C), the gradient of the scaled coordinates shared by q^ and k^.  ``f32_mfma`` selects the
    native f32 MFMA kernel instead of the split-bf16 products.  With ``coords`` (N, C) a fifth result
    d_sqrt_w (H, C) = sum_n dcs * coords is reduced in the same pass; rows at and after ``raw_size`` get zeros."""
    lib = _lib.load()
    if qhat.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("the backward pass needs f32 or bf16 tiles (rows of precision 'fp32' or 'bf16')")
    h, n, _ = qhat.shape
    tl = qpos.shape[0]
    gacc = _f32c(gacc, "grad of the partial sums")
    dev = qhat.device
    rows16 = qhat.dtype == torch.bfloat16
    part_dtype = torch.bfloat16 if rows16 else torch.float32
    dq_part = torch.empty(tl, n, h, 32, device=dev, dtype=part_dtype)
    dkv_part = torch.empty(tl, n, h, 64, device=dev, dtype=part_dtype)
    st = _stream(qhat)
    if rows16:   # the rows of the bf16 forward: one bf16 MFMA per product, bf16 per-table gradient rows (16-bit training)
        fn = lib.hept_block_attn_bwd_bf16
    else:
        fn = lib.hept_block_attn_bwd_f32mfma if f32_mfma else lib.hept_block_attn_bwd
    _lib.check(fn(qhat.data_ptr(), kvhat.data_ptr(), qpos.data_ptr(), kpos.data_ptr(), gacc.data_ptr(), n, h, head_dim,
                  tl, block_size, dq_part.data_ptr(), dkv_part.data_ptr(), st), "hept_block_attn_bwd")
    dq = torch.empty(n, h * head_dim, device=dev, dtype=torch.float32)
    dk, dv = torch.empty_like(dq), torch.empty_like(dq)
    dcs = torch.empty(n, h, coords_dim, device=dev, dtype=torch.float32)
    dsw = None
    if coords is not None:
        coords = _f32c(coords, "coords")
        dsw = torch.empty(h, coords_dim, device=dev, dtype=torch.float32)
    _lib.check((lib.hept_bwd_reduce16 if rows16 else lib.hept_bwd_reduce)(dq_part.data_ptr(), dkv_part.data_ptr(), tl, n, h, head_dim, coords_dim,
                                   coords.data_ptr() if coords is not None else None,
                                   n if raw_size is None else int(raw_size), dq.data_ptr(), dk.data_ptr(),
                                   dv.data_ptr(), dcs.data_ptr(), dsw.data_ptr() if dsw is not None else None, st),
               "hept_bwd_reduce")
    if coords is not None:
        return dq, dk, dv, dcs, dsw
    return dq, dk, dv, dcs


@_on_device
def combine_bwd(acc: torch.Tensor, g_out: torch.Tensor, out_weight: torch.Tensor, need_bias: bool = True):
    """Backward of ``combine_out`` on table-summed f32 rows: returns (gacc (N,H,32), d_weight (D,H*D), d_bias (D))."""
    lib = _lib.load()
    acc = _f32c(acc, "acc")
    g_out = _f32c(g_out, "grad of the output")
    w = _f32c(out_weight, "out_linear.weight")
    n, h, _ = acc.shape
    d = g_out.shape[1]
    gacc = torch.empty_like(acc)
    dw = torch.empty_like(w)
    db = torch.empty(d, device=acc.device, dtype=torch.float32) if need_bias else None
    scratch = torch.empty(int(lib.hept_combine_bwd_scratch_bytes_shape(n, h, d)), device=acc.device, dtype=torch.uint8)
    _lib.check(lib.hept_combine_bwd(acc.data_ptr(), g_out.data_ptr(), w.data_ptr(), n, h, d, gacc.data_ptr(),
                                    dw.data_ptr(), db.data_ptr() if db is not None else None, scratch.data_ptr(),
                                    scratch.numel(), _stream(acc)), "hept_combine_bwd")
    return gacc, dw, db


def geo_args(region_indices, regions_h, n_tables: int, n_heads: int, n: int):
    """The src variant's caller tensors in the C ABI's layout: eta, phi (T,H,N) f32 and cfac (T,H) f32
    (= ``ceil(regions_h[0]) + 1``, reference ``src/models/attention/hept.py:53``)."""
    eta, phi = region_indices
    eta = _f32c(eta, "region_indices[0]")
    phi = _f32c(phi, "region_indices[1]")
    if tuple(eta.shape) != (n_tables * n_heads, n) or phi.shape != eta.shape:
        raise ValueError(f"region_indices must be two {(n_tables * n_heads, n)} tensors, got {tuple(eta.shape)} "
                         f"and {tuple(phi.shape)}")
    rh = _f32c(regions_h, "regions_h")
    if tuple(rh.shape) != (2, n_tables * n_heads):
        raise ValueError(f"regions_h must have shape {(2, n_tables * n_heads)}, got {tuple(rh.shape)}")
    cfac = (torch.ceil(rh[0]) + 1).contiguous()
    return eta, phi, cfac


def _prepare(q, k, v, coords, codes, w_rpe_weight, alpha, block_size, w_per_dist):
    q, k, v, coords, w, alpha = (_f32c(x, nm) for x, nm in
                                 ((q, "query"), (k, "key"), (v, "value"), (coords, "coords"),
                                  (w_rpe_weight, "w_rpe.weight"), (alpha, "e2lsh.alpha")))
    n, h, d, c, t = _dims(q, coords, alpha)
    if k.shape != q.shape or v.shape != q.shape or coords.shape[0] != n:
        raise ValueError("query, key, value and coords must agree on the number of points")
    if n % block_size != 0:
        raise ValueError(f"number of points {n} is not a multiple of block_size {block_size}")
    if codes is not None and (codes.dtype != torch.int64 or not codes.is_cuda or tuple(codes.shape) != (t, h, n)):
        raise ValueError(f"combined_shifts must be an int64 GPU tensor of shape {(t, h, n)}, got {codes.dtype} {tuple(codes.shape)}")
    if w_per_dist == 0:  # precomputed sqrt_w (H, C) in place of the weight (hept_hip.h: K == 0)
        if w.shape != (h, c):
            raise ValueError(f"with w_per_dist=0 the weight argument is sqrt_w of shape {(h, c)}, got {tuple(w.shape)}")
    elif w.shape != (h * d, (c - 1) * w_per_dist):
        raise ValueError(f"w_rpe.weight must have shape {(h * d, (c - 1) * w_per_dist)}, got {tuple(w.shape)}")
    return q, k, v, coords, codes.contiguous() if codes is not None else None, w, alpha, (n, h, d, c, t)


@_on_device
def forward(q, k, v, coords, codes, w_rpe_weight, alpha, out_weight, out_bias, *, block_size: int, w_per_dist: int,
            precision="fp32", workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """Whole operator (reference ``example/hept.py:43-81``) in one C call; returns (N, D) float32."""
    lib = _lib.load()
    q, k, v, coords, codes, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, codes, w_rpe_weight, alpha,
                                                                block_size, w_per_dist)
    prec = precision_code(precision)
    _lib.check(_check_shape_cached(n, h, d, c, t, block_size), "hept_check_shape")
    need = _workspace_bytes_cached(n, h, d, c, t, block_size, prec)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=q.device, dtype=torch.uint8)
    ow = _f32c(out_weight, "out_linear.weight")
    ob = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    out = torch.empty(n, d, device=q.device, dtype=torch.float32)
    _lib.check(lib.hept_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), codes.data_ptr(),
                                w.data_ptr(), alpha.data_ptr(), ow.data_ptr(), ob.data_ptr() if ob is not None else None,
                                n, h, d, c, w_per_dist, t, block_size, prec, workspace.data_ptr(), workspace.numel(),
                                out.data_ptr(), stream if stream is not None else _stream(q)), "hept_forward")
    return out


def packed_partials(precision, head_dim: int) -> bool:
    """True if (precision, head_dim) produces packed 64-B partial rows (16-bit tiles, head_dim 24)."""
    return _lib.load().hept_part_precision(precision_code(precision), head_dim) == PREC_BF16


def _acc_buffer(n, h, packed, device):
    return torch.empty(n, h, 16 if packed else 32, device=device, dtype=torch.int32 if packed else torch.float32)


@_on_device
def forward_partial(q, k, v, coords, codes, w_rpe_weight, alpha, *, block_size: int, w_per_dist: int, t0: int,
                    tl: int, precision="fp32", workspace: Optional[torch.Tensor] = None,
                    packed: bool = False) -> torch.Tensor:
    """Tables [t0, t0+tl) only: returns acc (N, H, 32) = sum over those tables of [numer | denom]; ``packed``:
    the same sum as packed rows (N, H, 16) int32 (needs ``packed_partials(precision, D)``)."""
    lib = _lib.load()
    q, k, v, coords, codes, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, codes, w_rpe_weight, alpha,
                                                                block_size, w_per_dist)
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, tl, block_size), "hept_check_shape")
    need = int(lib.hept_workspace_bytes(n, h, d, c, tl, block_size, prec))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=q.device, dtype=torch.uint8)
    acc = _acc_buffer(n, h, packed, q.device)
    _lib.check(lib.hept_forward_partial(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), codes.data_ptr(),
                                        w.data_ptr(), alpha.data_ptr(), n, h, d, c, w_per_dist, t, t0, tl, block_size,
                                        prec, PREC_BF16 if packed else PREC_F32, workspace.data_ptr(),
                                        workspace.numel(), acc.data_ptr(), _stream(q)), "hept_forward_partial")
    return acc


@_on_device
def forward_src(q, k, v, coords, region_indices, regions_h, raw_size: int, w_rpe_weight, alpha, out_weight, out_bias,
                *, block_size: int, w_per_dist: int, precision="fp32",
                workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Whole operator of the reference's src variant (``src/models/attention/hept.py:74-117``); (N, D) float32."""
    lib = _lib.load()
    q, k, v, coords, _, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, None, w_rpe_weight, alpha, block_size,
                                                             w_per_dist)
    eta, phi, cfac = geo_args(region_indices, regions_h, t, h, n)
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, t, block_size), "hept_check_shape")
    need = int(lib.hept_workspace_bytes(n, h, d, c, t, block_size, prec))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=q.device, dtype=torch.uint8)
    ow = _f32c(out_weight, "out_linear.weight")
    ob = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    out = torch.empty(n, d, device=q.device, dtype=torch.float32)
    _lib.check(lib.hept_forward_src(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), eta.data_ptr(),
                                    phi.data_ptr(), cfac.data_ptr(), int(raw_size), w.data_ptr(), alpha.data_ptr(),
                                    ow.data_ptr(), ob.data_ptr() if ob is not None else None, n, h, d, c, w_per_dist,
                                    t, block_size, prec, workspace.data_ptr(), workspace.numel(), out.data_ptr(),
                                    _stream(q)), "hept_forward_src")
    return out


@_on_device
def forward_partial_src(q, k, v, coords, region_indices, regions_h, raw_size: int, w_rpe_weight, alpha, *,
                        block_size: int, w_per_dist: int, t0: int, tl: int, precision="fp32",
                        workspace: Optional[torch.Tensor] = None, packed: bool = False) -> torch.Tensor:
    """src variant, tables [t0, t0+tl) only: acc (N, H, 32) = sum over those tables of [numer | denom]."""
    lib = _lib.load()
    q, k, v, coords, _, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, None, w_rpe_weight, alpha, block_size,
                                                             w_per_dist)
    eta, phi, cfac = geo_args(region_indices, regions_h, t, h, n)
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, tl, block_size), "hept_check_shape")
    need = int(lib.hept_workspace_bytes(n, h, d, c, tl, block_size, prec))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=q.device, dtype=torch.uint8)
    acc = _acc_buffer(n, h, packed, q.device)
    _lib.check(lib.hept_forward_partial_src(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(),
                                            eta.data_ptr(), phi.data_ptr(), cfac.data_ptr(), int(raw_size),
                                            w.data_ptr(), alpha.data_ptr(), n, h, d, c, w_per_dist, t, t0, tl,
                                            block_size, prec, PREC_BF16 if packed else PREC_F32,
                                            workspace.data_ptr(), workspace.numel(), acc.data_ptr(), _stream(q)),
               "hept_forward_partial_src")
    return acc


@_on_device
def prep_hash_fused(x, norm_w, norm_b, eps, w_q, w_k, w_v, coords, sqrt_w, alpha, codes, precision="fp32",
                    t0: int = 0, tl: Optional[int] = None, raw_size: Optional[int] = None,
                    rows: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """``prep_hash`` with LayerNorm and the q/k/v projections fused in: ``x`` is the (N, D) input of the Attn block
    (reference ``example/transformer.py:155-156``); same outputs as :func:`prep_hash`."""
    lib = _lib.load()
    x, norm_w, norm_b, w_q, w_k, w_v, coords, sqrt_w, alpha = (
        _f32c(t_, nm) for t_, nm in ((x, "x"), (norm_w, "norm1.weight"), (norm_b, "norm1.bias"), (w_q, "w_q.weight"),
                                     (w_k, "w_k.weight"), (w_v, "w_v.weight"), (coords, "coords"),
                                     (sqrt_w, "sqrt_w"), (alpha, "e2lsh.alpha")))
    n, d = x.shape
    h, e, t = alpha.shape
    c = e - d
    if tuple(w_q.shape) != (h * d, d) or w_k.shape != w_q.shape or w_v.shape != w_q.shape:
        raise ValueError(f"w_q/w_k/w_v must have shape {(h * d, d)}")
    if coords.shape != (n, c) or norm_w.numel() != d or norm_b.numel() != d:
        raise ValueError("coords / norm1 parameters do not match x")
    if tuple(sqrt_w.shape) != (h, c):
        raise ValueError(f"sqrt_w must have shape {(h, c)}, got {tuple(sqrt_w.shape)}")
    if codes is not None:
        if codes.dtype != torch.int64 or tuple(codes.shape) != (t, h, n):
            raise ValueError(f"combined_shifts must be an int64 tensor of shape {(t, h, n)}")
        codes = codes.contiguous()
    raw_size = n if raw_size is None else int(raw_size)
    tl = t - t0 if tl is None else tl
    prec = precision_code(precision)
    dt = {PREC_F32: torch.float32, PREC_BF16: torch.bfloat16, PREC_MIXED16: torch.float16,
            PREC_F32_MFMA: torch.float32, PREC_F32_DIFF: torch.float32}[prec]
    dev = x.device
    if rows is not None:   # see prep_hash
        qhat, kvhat = rows
        if qhat.dtype != dt or tuple(qhat.shape) != (h, n, 32) or kvhat.dtype != dt or tuple(kvhat.shape) != (h, n, 64):
            raise ValueError("rows: buffers of another shape or precision")
    else:
        qhat = torch.empty(h, n, 32, device=dev, dtype=dt)
        kvhat = torch.empty(h, n, 64, device=dev, dtype=dt)
    qproj = torch.empty(tl, h, n, device=dev, dtype=torch.float32)
    kproj = torch.empty(tl, h, n, device=dev, dtype=torch.float32)
    minmax = torch.empty(tl, h, _lib.PREP_GRID, 4, device=dev, dtype=torch.float32)
    _lib.check(lib.hept_prep_hash_fused(x.data_ptr(), norm_w.data_ptr(), norm_b.data_ptr(), float(eps), w_q.data_ptr(),
                                        w_k.data_ptr(), w_v.data_ptr(), coords.data_ptr(), sqrt_w.data_ptr(),
                                        alpha.data_ptr(), codes.data_ptr() if codes is not None else None, n, raw_size,
                                        h, d, c, t, t0, tl, prec, qhat.data_ptr(), kvhat.data_ptr(), qproj.data_ptr(),
                                        kproj.data_ptr(), minmax.data_ptr(), _stream(x)), "hept_prep_hash_fused")
    return {"qhat": qhat, "kvhat": kvhat, "qproj": qproj, "kproj": kproj, "minmax": minmax}


@_on_device
def combine_ffn(part: torch.Tensor, head_dim: int, out_weight, out_bias, x, norm_w, norm_b, eps, ff1_w, ff1_b, ff2_w,
                ff2_b, n0: int = 0, n_count: Optional[int] = None) -> torch.Tensor:
    """``combine_out`` followed by the rest of the Attn block (residual, norm2, feed-forward, residual) in the
    same kernel; reference ``example/transformer.py:161-165`` in eval mode.  ``x`` is the full (N, D) block input."""
    lib = _lib.load()
    if part.dim() == 3:
        part = part.unsqueeze(0)
    tl, n, h, _ = part.shape
    n_count = n - n0 if n_count is None else n_count
    ts = [_f32c(t_, nm) for t_, nm in ((out_weight, "out_linear.weight"), (x, "x"), (norm_w, "norm2.weight"),
                                       (norm_b, "norm2.bias"), (ff1_w, "ff.0.weight"), (ff1_b, "ff.0.bias"),
                                       (ff2_w, "ff.2.weight"), (ff2_b, "ff.2.bias"))]
    ow, x, nw, nb, w1, b1, w2, b2 = ts
    ob = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    y = torch.empty(n_count, head_dim, device=part.device, dtype=torch.float32)
    _lib.check(lib.hept_combine_ffn(part.data_ptr(), _part_prec(part), tl, n, h, head_dim, n0, n_count, ow.data_ptr(),
                                    ob.data_ptr() if ob is not None else None, x[n0:].data_ptr(), nw.data_ptr(),
                                    nb.data_ptr(), float(eps), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                    b2.data_ptr(), y.data_ptr(), _stream(part)), "hept_combine_ffn")
    return y


@_on_device
def attn_block_forward(x, coords, codes, params: Dict[str, torch.Tensor], *, num_heads: int, block_size: int,
                       w_per_dist: int, eps1: float = 1e-5, eps2: float = 1e-5, precision="fp32",
                       workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The whole Attn block (reference ``example/transformer.py:154-165``, eval mode) in one C call.
    ``params`` holds the block's tensors under the reference's state-dict names."""
    lib = _lib.load()
    names = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "w_q": "w_q.weight", "w_k": "w_k.weight",
             "w_v": "w_v.weight", "w_rpe": "w_rpe.weight", "alpha": "attn.e2lsh.alpha",
             "out_w": "attn.out_linear.weight", "out_b": "attn.out_linear.bias", "norm2_w": "norm2.weight",
             "norm2_b": "norm2.bias", "ff1_w": "ff.0.weight", "ff1_b": "ff.0.bias", "ff2_w": "ff.2.weight",
             "ff2_b": "ff.2.bias"}
    x = _f32c(x, "x")
    coords = _f32c(coords, "coords")
    keep = {f: _f32c(params[k], k) for f, k in names.items()}
    n, d = x.shape
    h = num_heads
    e, t = keep["alpha"].shape[1], keep["alpha"].shape[2]
    c = e - d
    if n % block_size != 0:
        raise ValueError(f"number of points {n} is not a multiple of block_size {block_size}")
    if codes.dtype != torch.int64 or not codes.is_cuda or tuple(codes.shape) != (t, h, n):
        raise ValueError(f"combined_shifts must be an int64 GPU tensor of shape {(t, h, n)}")
    # the row builder indexes these by the sizes above: a mismatched tensor would be an out-of-bounds device read
    if keep["alpha"].shape[0] != h or tuple(coords.shape) != (n, c):
        raise ValueError(f"attn.e2lsh.alpha must be ({h}, D + C, n_hashes) and coords ({n}, {c})")
    want_rpe = (h * d, (c - 1) * w_per_dist) if w_per_dist > 0 else (h, c)
    if tuple(keep["w_rpe"].shape) != want_rpe:
        raise ValueError(f"w_rpe.weight must have shape {want_rpe}, got {tuple(keep['w_rpe'].shape)}")
    for f_ in ("w_q", "w_k", "w_v"):
        if tuple(keep[f_].shape) != (h * d, d):
            raise ValueError(f"{names[f_]} must have shape {(h * d, d)}")
    codes = codes.contiguous()
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, t, block_size), "hept_check_shape")
    need = int(lib.hept_workspace_bytes(n, h, d, c, t, block_size, prec))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.uint8)
    st = _lib.AttnParams(**{f: v.data_ptr() for f, v in keep.items()}, eps1=float(eps1), eps2=float(eps2))
    y = torch.empty(n, d, device=x.device, dtype=torch.float32)
    import ctypes

    _lib.check(lib.hept_attn_block_forward(x.data_ptr(), coords.data_ptr(), codes.data_ptr(), ctypes.byref(st), n, h,
                                           d, c, w_per_dist, t, block_size, prec, workspace.data_ptr(),
                                           workspace.numel(), y.data_ptr(), _stream(x)), "hept_attn_block_forward")
    return y


@_on_device
def partial_begin(q, k, v, coords, codes, w_rpe_weight, alpha, *, block_size: int, w_per_dist: int, t0: int, tl: int,
                  precision="fp32", workspace: torch.Tensor, geo=None) -> Tuple[int, int, int, int]:
    """Table sharding, first half (``hept_partial_begin``): parameter math, rows + hashes and the sort for tables
    [t0, t0+tl); everything stays in ``workspace`` for :func:`partial_heads`.  ``geo`` = (region_indices, regions_h,
    raw_size) selects the src variant (``codes`` is then None).  Returns (N, H, D, C)."""
    lib = _lib.load()
    q, k, v, coords, codes, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, codes, w_rpe_weight, alpha,
                                                                block_size, w_per_dist)
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, tl, block_size), "hept_check_shape")
    if workspace.numel() < int(lib.hept_workspace_bytes(n, h, d, c, tl, block_size, prec)):
        raise ValueError("workspace too small: size it with ops.workspace_bytes")
    if geo is None:
        rc = lib.hept_partial_begin(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), codes.data_ptr(),
                                    w.data_ptr(), alpha.data_ptr(), n, h, d, c, w_per_dist, t, t0, tl, block_size, prec,
                                    workspace.data_ptr(), workspace.numel(), _stream(q))
    else:
        region_indices, regions_h, raw_size = geo
        eta, phi, cfac = geo_args(region_indices, regions_h, t, h, n)
        rc = lib.hept_partial_begin_src(q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(), eta.data_ptr(),
                                        phi.data_ptr(), cfac.data_ptr(), int(raw_size), w.data_ptr(), alpha.data_ptr(),
                                        n, h, d, c, w_per_dist, t, t0, tl, block_size, prec, workspace.data_ptr(),
                                        workspace.numel(), _stream(q))
    _lib.check(rc, "hept_partial_begin")
    return n, h, d, c


@_on_device
def partial_heads(workspace: torch.Tensor, dims: Tuple[int, int, int, int], tl: int, block_size: int, precision,
                  h0: int, dst: torch.Tensor) -> None:
    """Second half (``hept_partial_heads``): block attention of heads [h0, h0 + dst.shape[1]) for the tables of the
    preceding :func:`partial_begin`, summed over those tables into ``dst`` (n_pad, hg, row) -- int32 rows of 16 (packed)
    or float32 rows of 32; rows at and after N are zero."""
    lib = _lib.load()
    n, h, d, c = dims
    n_pad, hg, row = dst.shape
    packed = dst.dtype == torch.int32
    if (packed and row != 16) or (not packed and (dst.dtype != torch.float32 or row != 32)) or not dst.is_contiguous():
        raise ValueError("dst must be a contiguous (n_pad, hg, 16) int32 or (n_pad, hg, 32) float32 tensor")
    _lib.check(lib.hept_partial_heads(workspace.data_ptr(), workspace.numel(), n, h, d, c, tl, block_size,
                                      precision_code(precision), h0, hg, n_pad, PREC_BF16 if packed else PREC_F32,
                                      dst.data_ptr(), _stream(dst)), "hept_partial_heads")


@_on_device
def combine_groups(part: torch.Tensor, head_dim: int, out_weight, out_bias, n0: int = 0,
                   n_count: Optional[int] = None) -> torch.Tensor:
    """``combine_out`` on rows that arrive split by head groups: ``part`` is (G, Tl, N, HG, row) -- group g holds heads
    [g*HG, (g+1)*HG) -- and is summed over Tl (the slices received from the ranks); returns (n_count, D)."""
    lib = _lib.load()
    if part.dim() != 5 or not part.is_contiguous():
        raise ValueError("part must be a contiguous (groups, tables, points, heads per group, row) tensor")
    g, tl, n, hg, row = part.shape
    n_count = n - n0 if n_count is None else n_count
    w = _f32c(out_weight, "out_linear.weight")
    b = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    out = torch.empty(n_count, head_dim, device=part.device, dtype=torch.float32)
    _lib.check(lib.hept_combine_groups(part.data_ptr(), _part_prec(part), tl, n, g * hg, head_dim, n0, n_count, hg,
                                       tl * n * hg * row, w.data_ptr(), b.data_ptr() if b is not None else None,
                                       out.data_ptr(), _stream(part)), "hept_combine_groups")
    return out


@_on_device
def forward_sharded(q, k, v, coords, codes, w_rpe_weight, alpha, out_weight, out_bias, *, comm: int, world: int,
                    block_size: int, w_per_dist: int, t0: int, tl: int, head_groups: int, precision="fp32",
                    workspace: torch.Tensor, xbuf: Optional[torch.Tensor] = None, one_sided: bool = False,
                    geo=None, out_view: bool = False, view_owner=None) -> torch.Tensor:
    """Table-sharded operator in one C call (``hept_forward_sharded``): this rank's tables [t0, t0+tl), the RCCL
    exchange pipelined by head groups on the communicator's side stream, combine of this rank's points and the
    all-gather; returns the full (N, D) output.  ``comm`` is a ``hept_comm*`` (see ``hept_amd.sharding``);
    ``one_sided`` selects the transport that stores rows straight into the peers' mapped exchange buffers
    (``hept_comm_p2p_*``; ``xbuf`` is then not needed).  ``out_view`` (one-sided transport, the communicator in view
    mode -- ``hept_comm_set_out_view``): the result is a tensor over the communicator's exchange buffer instead of a
    copy; it is overwritten by the SECOND next sharded call on this communicator (``TableSharding(out_view=True)``)."""
    lib = _lib.load()
    q, k, v, coords, codes, w, alpha, (n, h, d, c, t) = _prepare(q, k, v, coords, codes, w_rpe_weight, alpha,
                                                                block_size, w_per_dist)
    prec = precision_code(precision)
    _lib.check(lib.hept_check_shape(n, h, d, c, tl, block_size), "hept_check_shape")
    if workspace.numel() < int(lib.hept_workspace_bytes(n, h, d, c, tl, block_size, prec)):
        raise ValueError("workspace too small: size it with ops.workspace_bytes")
    if not one_sided and (xbuf is None or xbuf.numel() < int(lib.hept_exchange_bytes(n, h, d, world, prec))):
        raise ValueError("exchange buffer too small: size it with hept_exchange_bytes")
    ow = _f32c(out_weight, "out_linear.weight")
    ob = _f32c(out_bias, "out_linear.bias") if out_bias is not None else None
    per = (n + world - 1) // world
    view = bool(out_view and one_sided)
    out_full = None if view else torch.empty(per * world, d, device=q.device, dtype=torch.float32)
    tail = (n, h, d, c, w_per_dist, t, t0, tl, block_size, prec, head_groups,
            _lib.TRANSPORT_ONE_SIDED if one_sided else _lib.TRANSPORT_RCCL, workspace.data_ptr(), workspace.numel(),
            xbuf.data_ptr() if xbuf is not None else None, xbuf.numel() if xbuf is not None else 0,
            out_full.data_ptr() if out_full is not None else None, _stream(q))
    if geo is None:
        rc = lib.hept_forward_sharded(comm, q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(),
                                      codes.data_ptr(), w.data_ptr(), alpha.data_ptr(), ow.data_ptr(),
                                      ob.data_ptr() if ob is not None else None, *tail)
    else:
        region_indices, regions_h, raw_size = geo
        eta, phi, cfac = geo_args(region_indices, regions_h, t, h, n)
        rc = lib.hept_forward_sharded_src(comm, q.data_ptr(), k.data_ptr(), v.data_ptr(), coords.data_ptr(),
                                          eta.data_ptr(), phi.data_ptr(), cfac.data_ptr(), int(raw_size), w.data_ptr(),
                                          alpha.data_ptr(), ow.data_ptr(), ob.data_ptr() if ob is not None else None,
                                          *tail)
    _lib.check(rc, "hept_forward_sharded")
    if view:
        import ctypes

        ptr = ctypes.c_void_p()
        _lib.check(lib.hept_comm_out_view(comm, ctypes.byref(ptr)), "hept_comm_out_view")
        return torch.as_tensor(_DeviceRows(ptr.value, n, d, view_owner), device=q.device)
    return out_full[:n]


class _DeviceRows:
    """(n, d) f32 rows at a device address owned by the C library (the exchange buffer of a communicator), for
    ``torch.as_tensor`` (zero-copy through ``__cuda_array_interface__``)"""

    def __init__(self, ptr: int, n: int, d: int, owner=None):
        self.__cuda_array_interface__ = {"shape": (n, d), "typestr": "<f4", "data": (ptr, False), "version": 2,
                                         "strides": None}
        # the tensor made from this object keeps it alive, and with it the owner of the memory (the TableSharding whose
        # communicator holds the exchange buffer): the buffer cannot be unmapped under a live view
        self.owner = owner
        if owner is not None:
            owner._live_views = getattr(owner, "_live_views", 0) + 1

    def __del__(self):
        if self.owner is not None:
            self.owner._live_views = max(0, getattr(self.owner, "_live_views", 1) - 1)


# ---- the dense tail of the Attn block's training step (csrc/block_train.hip; rows of 24 floats)
@_on_device
def rows_wgrad(d_y: torch.Tensor, x: torch.Tensor, need_bias: bool = False):
    """Weight gradient of a ``Linear(24 -> O)``: ``d_y.t() @ x`` (O, 24) (and the column sums of ``d_y`` for the bias)
    with a fixed-order two-stage reduction over the points -- rocBLAS runs this shape at ~140 us for 60k points."""
    lib = _lib.load()
    d_y, x = _f32c(d_y, "d_y"), _f32c(x, "x")
    n, o = d_y.shape
    if x.shape != (n, 24):
        raise ValueError(f"x must be ({n}, 24), got {tuple(x.shape)}")
    dw = torch.empty(o, 24, device=x.device, dtype=torch.float32)
    db = torch.empty(o, device=x.device, dtype=torch.float32) if need_bias else None
    scratch = torch.empty(int(lib.hept_rows_wgrad_scratch_bytes(n, o)), device=x.device, dtype=torch.uint8)
    _lib.check(lib.hept_rows_wgrad(d_y.data_ptr(), x.data_ptr(), n, o, 24, dw.data_ptr(),
                                   db.data_ptr() if db is not None else None, scratch.data_ptr(), scratch.numel(),
                                   _stream(x)), "hept_rows_wgrad")
    return (dw, db) if need_bias else dw


@_on_device
def ln_bwd(x: torch.Tensor, d_xn: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float):
    """LayerNorm(24) backward: returns (dx, xn, d_ln_w, d_ln_b) -- ``xn`` = the normalised rows (recomputed)."""
    lib = _lib.load()
    x, d_xn, ln_w, ln_b = (_f32c(t, nm) for t, nm in ((x, "x"), (d_xn, "d_xn"), (ln_w, "ln.weight"), (ln_b, "ln.bias")))
    n, d = x.shape
    dx, xn = torch.empty_like(x), torch.empty_like(x)
    dlw, dlb = torch.empty_like(ln_w), torch.empty_like(ln_b)
    scratch = torch.empty(int(lib.hept_ln_scratch_bytes(n)), device=x.device, dtype=torch.uint8)
    _lib.check(lib.hept_ln_bwd(x.data_ptr(), d_xn.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), n, d,
                               dx.data_ptr(), xn.data_ptr(), dlw.data_ptr(), dlb.data_ptr(), scratch.data_ptr(),
                               scratch.numel(), _stream(x)), "hept_ln_bwd")
    return dx, xn, dlw, dlb


@_on_device
def ln_ffn_fwd(x1, ln_w, ln_b, eps, w1, b1, w2, b2) -> torch.Tensor:
    """``ff.2(relu(ff.0(norm2(x1))))`` (reference ``example/transformer.py:162``) in one kernel; (N, 24)."""
    lib = _lib.load()
    ts = [_f32c(t, nm) for t, nm in ((x1, "x1"), (ln_w, "norm2.weight"), (ln_b, "norm2.bias"), (w1, "ff.0.weight"),
                                     (b1, "ff.0.bias"), (w2, "ff.2.weight"), (b2, "ff.2.bias"))]
    x1, ln_w, ln_b, w1, b1, w2, b2 = ts
    n, d = x1.shape
    out = torch.empty_like(x1)
    _lib.check(lib.hept_ln_ffn_fwd(x1.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), w1.data_ptr(),
                                   b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), n, d, out.data_ptr(), _stream(x1)),
               "hept_ln_ffn_fwd")
    return out


@_on_device
def ln_ffn_bwd(x1, d_out, ln_w, ln_b, eps, w1, b1, w2, b2):
    """Backward of :func:`ln_ffn_fwd`: (d_x1, d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2)."""
    lib = _lib.load()
    ts = [_f32c(t, nm) for t, nm in ((x1, "x1"), (d_out, "d_out"), (ln_w, "norm2.weight"), (ln_b, "norm2.bias"),
                                     (w1, "ff.0.weight"), (b1, "ff.0.bias"), (w2, "ff.2.weight"), (b2, "ff.2.bias"))]
    x1, d_out, ln_w, ln_b, w1, b1, w2, b2 = ts
    n, d = x1.shape
    outs = [torch.empty_like(t) for t in (x1, ln_w, ln_b, w1, b1, w2, b2)]
    scratch = torch.empty(int(lib.hept_ln_ffn_bwd_scratch_bytes(n)), device=x1.device, dtype=torch.uint8)
    _lib.check(lib.hept_ln_ffn_bwd(x1.data_ptr(), d_out.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), float(eps),
                                   w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), n, d,
                                   *[t.data_ptr() for t in outs], scratch.data_ptr(), scratch.numel(), _stream(x1)),
               "hept_ln_ffn_bwd")
    return tuple(outs)


def profile_enable(mode: int, max_calls: int = 0, stride: int = 1) -> None:
    """Stage timing with HIP events recorded inside hept_forward (mode 1: block_attn only, 2: all stages, 0: off);
    only every ``stride``-th call is bracketed."""
    lib = _lib.load()
    _lib.check(lib.hept_profile_enable(mode, max_calls), "hept_profile_enable")
    _lib.check(lib.hept_profile_stride(stride), "hept_profile_stride")


def profile_stride(stride: int) -> None:
    """Bracket every ``stride``-th call from the next call on (restarts the count)."""
    _lib.check(_lib.load().hept_profile_stride(stride), "hept_profile_stride")


def profile_read() -> Tuple[Dict[str, float], int]:
    """Summed milliseconds per stage over the recorded calls, and the number of calls; resets the pool."""
    import ctypes

    ms = (ctypes.c_float * 7)()
    n = ctypes.c_int(0)
    _lib.check(_lib.load().hept_profile_read(ms, ctypes.byref(n)), "hept_profile_read")
    # (the sharded call: "combine" = the exposed push / transfer of the last head group, then its own two stages;
    #  "chunk_sort" = the first of the sort's two launches, inside "sort_tables")
    return dict(zip(("prep_hash", "sort_tables", "block_attn", "combine", "sharded_combine", "sharded_gather", "chunk_sort"),
                    [float(x) for x in ms])), int(n.value)
