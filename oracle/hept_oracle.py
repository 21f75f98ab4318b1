"""CPU oracle for the HEPT LSH block-attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product path (``hept_amd``) never imports this module and has no CPU fallback.

What it is: a stage-by-stage restatement, in eager CPU torch ops, of the
algorithm of the reference operator ``HEPTAttention.forward``
(reference ``example/hept.py:43-81`` with helpers ``example/hept_utils.py:38-97``).
Each function cites the reference lines it follows.  The arithmetic keeps the
reference's operation order (fp32, same ATen primitives for the matmuls, the
exp and the reductions) so that, in the same container, it reproduces the
imported reference bit for bit when given the same sort permutations.

Parity pinning: ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference/example``) in the build container and stores its inputs,
intermediates and outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this module against every one of them.
The reference ships no tests or known-answer vectors of its own.

Two deliberate, documented differences from the reference:

* ``sort_keys(..., stable=True)`` (default) uses a *stable* ascending sort.
  The reference calls ``argsort`` without ``stable=True``
  (``example/hept.py:67-68``); its tie order is implementation defined.  Every
  stage after the sort accepts injected permutations so that the reference's
  own permutation can be replayed exactly.
* ``tile_dtype`` lets the bucket attention round the gathered q̂/k̂/v tiles, the
  attention weights and the per-table numerators to bf16 (fp32 accumulate, fp32
  row norms of the *rounded* values, fp32 denominators) to model the bf16 MFMA
  path of the HIP kernel.  ``torch.float32`` reproduces the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

__all__ = [
    "rpe_scale",
    "augment_qk",
    "e2lsh_project",
    "hash_range",
    "shifted_keys",
    "geo_shift",
    "sort_keys",
    "gather_blocks",
    "block_rbf_attention",
    "unsort_tables",
    "combine_tables",
    "out_projection",
    "forward",
    "forward_partials",
    "attn_block",
    "HeptShapes",
]


@dataclass(frozen=True)
class HeptShapes:
    """Static sizes of one call (names as in SURVEY.md: N,H,D,C,E,T,B,K)."""

    n_points: int
    n_heads: int
    head_dim: int
    coords_dim: int
    n_tables: int
    block_size: int
    w_per_dist: int

    @property
    def hash_dim(self) -> int:
        return self.head_dim + self.coords_dim

    @property
    def n_blocks(self) -> int:
        return self.n_points // self.block_size


# ---------------------------------------------------------------------------
# stage a3 + a4: relative-position scale and coordinate augmentation
# ---------------------------------------------------------------------------
def rpe_scale(w_rpe_weight: torch.Tensor, n_heads: int, head_dim: int, w_per_dist: int) -> torch.Tensor:
    """Per-head, per-coordinate factor ``sqrt(2 * qw)`` of shape (H, C).

    Follows ``example/hept.py:48-54`` (the ``(h d) (r k) -> h d r k`` view of
    ``w_rpe.weight``) and ``example/hept.py:22-23,25``: sum over the head
    channels ``d``, clamp at 50, exp, sum over the ``k`` mixture weights, then
    column 0 is duplicated in front (eta and phi share the first distance
    weight) and ``sqrt(2 * .)`` is taken.
    """
    hd, rk = w_rpe_weight.shape
    assert hd == n_heads * head_dim and rk % w_per_dist == 0
    n_dist = rk // w_per_dist
    w4 = w_rpe_weight.reshape(n_heads, head_dim, n_dist, w_per_dist)
    qw = w4.sum(dim=1).clamp(max=50).exp().sum(dim=-1)  # (H, C-1)
    qw_full = torch.cat([qw[:, :1], qw], dim=-1)  # (H, C)
    return torch.sqrt(2 * qw_full)


def augment_qk(
    q: torch.Tensor, k: torch.Tensor, coords: torch.Tensor, sqrt_w: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """q̂ = [q ‖ sqrt_w·coords], k̂ = [k ‖ sqrt_w·coords]; returns (H, N, E) each.

    ``q``/``k`` are (N, H*D).  Follows ``example/hept.py:44-45`` (head view),
    ``:25-27`` (scaled coordinates appended) and ``:57-58`` (head-major view).
    """
    n_heads = sqrt_w.shape[0]
    n = q.shape[0]
    qh = q.reshape(n, n_heads, -1)
    kh = k.reshape(n, n_heads, -1)
    scaled = sqrt_w[None] * coords[:, None]  # (N, H, C)
    q_hat = torch.cat([qh, scaled], dim=-1).permute(1, 0, 2)
    k_hat = torch.cat([kh, scaled], dim=-1).permute(1, 0, 2)
    return q_hat, k_hat


# ---------------------------------------------------------------------------
# stage a2 + a6 + a7: E2LSH projection, hash range, AND-code shift
# ---------------------------------------------------------------------------
def e2lsh_project(x_hat: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """(H,N,E) x (H,E,T) -> (T,H,N) real-valued hash, ``example/hept_utils.py:45-47``."""
    return torch.bmm(x_hat, alpha).permute(2, 0, 1)


def hash_range(q_hashed: torch.Tensor, k_hashed: torch.Tensor) -> torch.Tensor:
    """max over points of (q,k) hashes minus min, shape (T,H,1); ``example/hept_utils.py:68-70``."""
    hi = torch.max(q_hashed.max(-1, keepdim=True).values, k_hashed.max(-1, keepdim=True).values)
    lo = torch.min(q_hashed.min(-1, keepdim=True).values, k_hashed.min(-1, keepdim=True).values)
    return hi - lo


def shifted_keys(
    q_hashed: torch.Tensor, k_hashed: torch.Tensor, codes: torch.Tensor, span: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sort keys ``hash + float(code) * span``: one rounded multiply, one rounded add.

    ``codes`` is the int64 (T,H,N) AND code (``combined_shifts``);
    ``example/hept.py:63-65``.
    """
    offs = codes * span  # int64 * fp32 -> fp32 (type promotion), rounded once
    return q_hashed + offs, k_hashed + offs


def geo_shift(regions_h: torch.Tensor, span: torch.Tensor, region_indices, n_tables: int) -> torch.Tensor:
    """Float region shift of the reference's ``src`` variant, shape (T,H,N); ``src/models/attention/hept.py:46-56``.

    ``region_indices`` = (eta, phi), each (T*H, N) float, rows ordered ``(c h)``; ``regions_h`` (2, T*H);
    ``span`` (T,H,1).  Operation order as the reference: ``eta*span``; ``(phi*span) * (ceil(regions_h[0]) + 1)``;
    then phi part + eta part.  The reference computes the same tensor twice (for q and for k).
    """
    eta, phi = region_indices
    span_f = span.reshape(-1, span.shape[-1])  # "c h d -> (c h) d", src/models/attention/hept.py:94
    shift_eta = eta * span_f
    shift_phi = phi * span_f * (torch.ceil(regions_h[0][:, None]) + 1)
    res = shift_phi + shift_eta
    return res.reshape(n_tables, -1, res.shape[-1])


def sort_keys(keys: torch.Tensor, stable: bool = True) -> torch.Tensor:
    """Ascending permutation along the point axis, ``example/hept.py:67-68``.

    ``stable=True`` (the oracle's contract, shared with the HIP radix sort):
    ties keep ascending point index.  ``stable=False`` is the reference call.
    """
    if stable:
        return torch.sort(keys, dim=-1, stable=True).indices
    return keys.argsort(dim=-1)


# ---------------------------------------------------------------------------
# stage a9 + a10: gather into blocks, block-local RBF attention
# ---------------------------------------------------------------------------
def gather_blocks(x: torch.Tensor, perm: torch.Tensor, block_size: int) -> torch.Tensor:
    """out[t,h,b,i,:] = x[h, perm[t,h,b*B+i], :]; ``example/hept_utils.py:74-92``."""
    t, h, n = perm.shape
    width = x.shape[-1]
    if n % block_size != 0:
        raise ValueError(f"number of points {n} is not a multiple of block_size {block_size}")
    idx = perm.unsqueeze(-1).expand(t, h, n, width)
    rows = x.unsqueeze(0).expand(t, h, n, width).gather(-2, idx)
    return rows.reshape(t, h, n // block_size, block_size, width)


def _round_tile(x: torch.Tensor, tile_dtype: torch.dtype) -> torch.Tensor:
    if tile_dtype == torch.float32 or tile_dtype == x.dtype:
        return x
    return x.to(tile_dtype).to(x.dtype)


def block_rbf_attention(
    sq: torch.Tensor, sk: torch.Tensor, sv: torch.Tensor, p_dtype: torch.dtype = torch.float32
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Un-normalised block attention, ``example/hept.py:7-18``.

    ``S = q·kᵀ - ½|q|² - ½|k|²ᵀ`` → ``exp(min(S, 0))``; returns
    ``(denom, numer)`` = (row sums + 1e-20, S·v).  No row max, no division.
    ``p_dtype=torch.bfloat16`` rounds the weights before the row sum and S·v
    (models the bf16 MFMA path; not reference behaviour).
    """
    qn = -0.5 * (sq**2).sum(dim=-1, keepdim=True)
    kn = -0.5 * (sk**2).sum(dim=-1, keepdim=True)
    s = torch.matmul(sq, sk.transpose(-1, -2))
    s = (s + qn + kn.transpose(-1, -2)).clamp(max=0.0).exp()
    s = _round_tile(s, p_dtype)
    denom = s.sum(dim=-1, keepdim=True) + 1e-20
    numer = torch.matmul(s, sv)
    return denom, numer


def unsort_tables(x_sorted: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """Back to original point order: out[t,h,perm[t,h,i],:] = x_sorted[t,h,i,:].

    Equivalent to ``invert_permutation`` + ``unsort_from_buckets``
    (``example/hept_utils.py:50-61,95-97``; calls ``example/hept.py:76-78``).
    """
    t, h, n = perm.shape
    flat = x_sorted.reshape(t, h, n, -1)
    out = torch.empty_like(flat)
    out.scatter_(-2, perm.unsqueeze(-1).expand_as(flat), flat)
    return out


# ---------------------------------------------------------------------------
# stage a13 + a14: OR-combine over tables, output projection
# ---------------------------------------------------------------------------
def combine_tables(numer: torch.Tensor, denom: torch.Tensor) -> torch.Tensor:
    """Σ_t numer / Σ_t denom → (H,N,D); ``example/hept.py:79``."""
    return numer.sum(dim=0) / denom.sum(dim=0)


def out_projection(per_head: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """(H,N,D) → (N,H*D) → Linear(H*D→D); ``example/hept.py:80``."""
    h, n, d = per_head.shape
    flat = per_head.permute(1, 0, 2).reshape(n, h * d)
    return torch.nn.functional.linear(flat, weight, bias)


# ---------------------------------------------------------------------------
# whole operator
# ---------------------------------------------------------------------------
def forward_partials(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    coords: torch.Tensor,
    codes: torch.Tensor,
    w_rpe_weight: torch.Tensor,
    alpha: torch.Tensor,
    *,
    block_size: int,
    w_per_dist: int,
    stable_sort: bool = True,
    q_positions: Optional[torch.Tensor] = None,
    k_positions: Optional[torch.Tensor] = None,
    tile_dtype: torch.dtype = torch.float32,
    qk_dtype: Optional[torch.dtype] = None,
    keep: bool = True,
    grad: bool = False,
    geo: Optional[Dict] = None,
) -> Dict[str, torch.Tensor]:
    """Everything up to the per-table partials in original point order.

    ``geo`` selects the reference's ``src`` variant (``src/models/attention/hept.py:74-117``): a dict with
    ``raw_size``, ``region_indices`` (eta, phi) and ``regions_h`` as built by the caller
    (``src/models/baselines/transformer.py:43-57``); ``codes`` is then unused (pass None).  Rows at and after
    ``raw_size`` are zeroed in q̂, k̂ and v (``:89-91``), still take part in the hash range with their zero
    hashes, and are then given the hash +inf (``:95-96``) so that they sort last.

    ``grad=True`` leaves autograd on (the reference trains through these very ops with plain
    autograd, ``example/trainer.py:11-22``; hashing stays ``no_grad`` as in
    ``example/hept_utils.py:64``), so ``out.backward()`` yields the reference's gradients.

    ``alpha``/``codes`` may hold a *subset* of the tables (table sharding,
    SURVEY.md §8e): every stage is independent per table until
    ``combine_tables``.  ``q_positions``/``k_positions`` inject permutations
    (skipping ``sort_keys``).
    """
    if not grad:
        with torch.no_grad():
            return forward_partials(q, k, v, coords, codes, w_rpe_weight, alpha, block_size=block_size,
                                    w_per_dist=w_per_dist, stable_sort=stable_sort, q_positions=q_positions,
                                    k_positions=k_positions, tile_dtype=tile_dtype, qk_dtype=qk_dtype, keep=keep,
                                    grad=True, geo=geo)
    n_heads, hash_dim, _ = alpha.shape
    head_dim = q.shape[1] // n_heads
    if q.shape[0] % block_size != 0:
        raise ValueError(f"number of points {q.shape[0]} is not a multiple of block_size {block_size}")
    sqrt_w = rpe_scale(w_rpe_weight, n_heads, head_dim, w_per_dist)
    q_hat, k_hat = augment_qk(q, k, coords, sqrt_w)
    assert q_hat.shape[-1] == hash_dim
    v_h = v.reshape(v.shape[0], n_heads, head_dim).permute(1, 0, 2)
    if geo is not None:
        # the reference's in-place zero fill of the padding rows (:89-91), written the same way so that autograd
        # records the same graph (gradient accumulation order included)
        raw = int(geo["raw_size"])
        if not v_h.requires_grad and not torch.is_grad_enabled():
            q_hat, k_hat, v_h = q_hat.clone(), k_hat.clone(), v_h.clone()
        elif v_h.is_leaf or v_h._base is not None and v_h._base.is_leaf:
            v_h = (v * 1.0).reshape(v.shape[0], n_heads, head_dim).permute(1, 0, 2)
        q_hat[:, raw:] = 0.0
        k_hat[:, raw:] = 0.0
        v_h[:, raw:] = 0.0

    with torch.no_grad():  # lsh_mapping is @torch.no_grad in the reference; argsort yields integers
        q_hashed = e2lsh_project(q_hat, alpha)
        k_hashed = e2lsh_project(k_hat, alpha)
        span = hash_range(q_hashed, k_hashed)
        if geo is None:
            q_keys, k_keys = shifted_keys(q_hashed, k_hashed, codes, span)
        else:
            q_hashed = q_hashed.clone()
            k_hashed = k_hashed.clone()
            q_hashed[..., int(geo["raw_size"]):] = float("inf")
            k_hashed[..., int(geo["raw_size"]):] = float("inf")
            shift = geo_shift(geo["regions_h"], span, geo["region_indices"], alpha.shape[-1])
            q_keys, k_keys = q_hashed + shift, k_hashed + shift
        q_pos = sort_keys(q_keys, stable_sort) if q_positions is None else q_positions
        k_pos = sort_keys(k_keys, stable_sort) if k_positions is None else k_positions

    qk_dt = tile_dtype if qk_dtype is None else qk_dtype  # "mixed16": fp16 q̂/k̂ tiles, everything else bf16
    sq = gather_blocks(_round_tile(q_hat, qk_dt), q_pos, block_size)
    sk = gather_blocks(_round_tile(k_hat, qk_dt), k_pos, block_size)
    sv = gather_blocks(_round_tile(v_h, tile_dtype), k_pos, block_size)
    denom_s, numer_s = block_rbf_attention(sq, sk, sv, tile_dtype)
    del sq, sk, sv
    # the bf16 HIP path stores each table's numerators as bf16 (denominators stay fp32)
    numer = unsort_tables(_round_tile(numer_s, tile_dtype), q_pos)
    denom = unsort_tables(denom_s, q_pos)
    res = {"numer": numer, "denom": denom, "q_positions": q_pos, "k_positions": k_pos}
    if keep:
        res.update(
            sqrt_w=sqrt_w,
            q_hat=q_hat,
            k_hat=k_hat,
            q_hashed=q_hashed,
            k_hashed=k_hashed,
            hash_span=span,
            q_keys=q_keys,
            k_keys=k_keys,
        )
    return res


def forward(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    coords: torch.Tensor,
    codes: torch.Tensor,
    w_rpe_weight: torch.Tensor,
    alpha: torch.Tensor,
    out_weight: torch.Tensor,
    out_bias: Optional[torch.Tensor],
    *,
    block_size: int,
    w_per_dist: int,
    stable_sort: bool = True,
    q_positions: Optional[torch.Tensor] = None,
    k_positions: Optional[torch.Tensor] = None,
    tile_dtype: torch.dtype = torch.float32,
    qk_dtype: Optional[torch.dtype] = None,
    keep: bool = True,
    grad: bool = False,
    geo: Optional[Dict] = None,
) -> Dict[str, torch.Tensor]:
    """Full operator, ``example/hept.py:43-81`` (``geo`` given: ``src/models/attention/hept.py:74-117``);
    returns a dict with ``out`` (N,D) and intermediates."""
    res = forward_partials(
        q, k, v, coords, codes, w_rpe_weight, alpha,
        block_size=block_size, w_per_dist=w_per_dist, stable_sort=stable_sort,
        q_positions=q_positions, k_positions=k_positions, tile_dtype=tile_dtype, qk_dtype=qk_dtype, keep=keep,
        grad=grad, geo=geo,
    )
    if not grad:
        with torch.no_grad():
            per_head = combine_tables(res["numer"], res["denom"])
            res["per_head"] = per_head
            res["out"] = out_projection(per_head, out_weight, out_bias)
        return res
    per_head = combine_tables(res["numer"], res["denom"])
    res["per_head"] = per_head
    res["out"] = out_projection(per_head, out_weight, out_bias)
    return res


def attn_block(
    x: torch.Tensor,
    coords: torch.Tensor,
    codes: torch.Tensor,
    params: Dict[str, torch.Tensor],
    *,
    num_heads: int,
    block_size: int,
    w_per_dist: int,
    eps: float = 1e-5,
    **kw,
) -> Dict[str, torch.Tensor]:
    """The ``Attn`` block around the operator in eval mode (dropout = identity), ``example/transformer.py:154-165``.

    ``params`` holds the block's tensors under the reference's state-dict names.  Returns ``y`` (N, D), the
    operator's output ``aggr`` and the operator's intermediates.
    """
    import torch.nn.functional as F

    d = x.shape[1]
    with torch.no_grad():
        x_normed = F.layer_norm(x, (d,), params["norm1.weight"], params["norm1.bias"], eps)      # :155
        q = F.linear(x_normed, params["w_q.weight"])                                             # :156
        k = F.linear(x_normed, params["w_k.weight"])
        v = F.linear(x_normed, params["w_v.weight"])
        res = forward(q, k, v, coords, codes, params["w_rpe.weight"], params["attn.e2lsh.alpha"],
                      params["attn.out_linear.weight"], params["attn.out_linear.bias"], block_size=block_size,
                      w_per_dist=w_per_dist, **kw)                                               # :157
        x1 = x + res["out"]                                                                      # :161
        hidden = F.relu(F.linear(F.layer_norm(x1, (d,), params["norm2.weight"], params["norm2.bias"], eps),
                                 params["ff.0.weight"], params["ff.0.bias"]))
        ff_output = F.linear(hidden, params["ff.2.weight"], params["ff.2.bias"])                 # :162
        res["aggr"] = res["out"]
        res["q"], res["k"], res["v"] = q, k, v
        res["y"] = x1 + ff_output                                                                # :163
    return res
