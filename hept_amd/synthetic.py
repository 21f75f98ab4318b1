"""Seeded synthetic point clouds of the tracking / pileup shapes.

Everything is drawn on the CPU from one ``torch.Generator`` so that the CPU
oracle and the GPU path see identical data; tensors are moved to ``device`` at
the end.  There is no dataset or checkpoint download in this build.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch

from .prep import get_regions, prepare_input, prepare_input_src

__all__ = ["make_inputs", "make_inputs_src", "WORKLOADS", "workload_inputs"]


def _linear_init(gen: torch.Generator, out_f: int, in_f: int, bias: bool):
    """nn.Linear's default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound if bias else None
    return w, b


def make_inputs(
    cloud_sizes: Sequence[int],
    *,
    block_size: int,
    n_hashes: int,
    coords_dim: int = 6,
    num_heads: int = 8,
    h_dim: int = 24,
    num_regions: int = 150,
    num_w_per_dist: int = 10,
    seed: int = 0,
    device: str = "cpu",
    regions: Optional[torch.Tensor] = None,
    pad_seq: Optional[torch.Tensor] = None,
    cluster_size: int = 0,
    cluster_spread: float = 0.05,
) -> Dict[str, torch.Tensor]:
    """One batch of clouds, padded per cloud to a multiple of ``block_size``.

    q, k, v ~ N(0,1) per *raw* point and then gathered with ``pad_seq`` (pad
    slots replicate real points, as in the reference model where q = W_q(x[pad_seq])).
    ``regions`` / ``pad_seq`` can be injected (golden fixtures replay the
    reference's own values).  ``cluster_size > 0`` draws "track-like" data:
    groups of that many points share a centre in feature and coordinate space
    (plus ``cluster_spread``·N(0,1)), scattered over the cloud in random order,
    so that points hashed into one block carry O(1) attention weight.
    """
    gen = torch.Generator().manual_seed(seed)
    n_raw = int(sum(cloud_sizes))
    hd = num_heads * h_dim
    q = torch.randn(n_raw, hd, generator=gen)
    k = torch.randn(n_raw, hd, generator=gen)
    v = torch.randn(n_raw, hd, generator=gen)
    coords = torch.randn(n_raw, coords_dim, generator=gen)
    cluster_id = None
    if cluster_size > 0:
        n_clu = (n_raw + cluster_size - 1) // cluster_size
        cluster_id = (torch.randperm(n_raw, generator=gen) // cluster_size).clamp(max=n_clu - 1)
        centres = torch.randn(n_clu, 3 * hd + coords_dim, generator=gen)[cluster_id]
        q = centres[:, :hd] + cluster_spread * q
        k = centres[:, :hd] + cluster_spread * k
        v = centres[:, 2 * hd : 3 * hd] + cluster_spread * v
        coords = centres[:, 3 * hd :] + cluster_spread * coords
    w_rpe, _ = _linear_init(gen, hd, (coords_dim - 1) * num_w_per_dist, bias=False)
    alpha = torch.randn(num_heads, h_dim + coords_dim, n_hashes, generator=gen)
    out_w, out_b = _linear_init(gen, h_dim, hd, bias=True)
    if regions is None:
        regions = get_regions(num_regions, n_hashes, num_heads, generator=gen)
    batch = torch.repeat_interleave(torch.arange(len(cloud_sizes)), torch.tensor(list(cloud_sizes)))

    helper = {"block_size": block_size, "num_heads": num_heads, "regions": regions}
    if pad_seq is None:
        # x = arange, so the returned "padded features" are the gather index itself
        pad_seq_t, kw, unpad = prepare_input(torch.arange(n_raw), coords, batch, helper)
        codes, coords_p = kw["combined_shifts"], kw["coords"]
    else:
        # codes of the raw points (block_size 1 = no padding), then the injected padding
        _, kw, _ = prepare_input(torch.arange(n_raw), coords, batch, {**helper, "block_size": 1})
        pad_seq_t = pad_seq.long()
        codes, coords_p = kw["combined_shifts"][..., pad_seq_t], coords[pad_seq_t]
        unpad = torch.ones(pad_seq_t.numel(), dtype=torch.bool)
        sizes = torch.tensor(list(cloud_sizes))
        padded = ((sizes + block_size - 1) // block_size) * block_size
        ends = padded.cumsum(0)
        for i in range(len(cloud_sizes)):
            unpad[int(ends[i] - (padded[i] - sizes[i])) : int(ends[i])] = False

    out = {
        "q": q[pad_seq_t].contiguous(),
        "k": k[pad_seq_t].contiguous(),
        "v": v[pad_seq_t].contiguous(),
        "coords": coords_p.contiguous(),
        "combined_shifts": codes.contiguous(),
        "w_rpe_weight": w_rpe,
        "alpha": alpha,
        "out_weight": out_w,
        "out_bias": out_b,
        "regions": regions,
        "unpad_seq": unpad,
        "pad_seq": pad_seq_t,
        "batch": batch,
        "coords_raw": coords,
    }
    if cluster_id is not None:
        out["cluster_id"] = cluster_id
    out = {name: t.to(device) for name, t in out.items()}
    out["n_raw"] = n_raw
    return out


def make_inputs_src(
    raw_size: int,
    *,
    block_size: int,
    n_hashes: int,
    coords_dim: int = 6,
    num_heads: int = 8,
    h_dim: int = 24,
    num_regions: int = 150,
    num_w_per_dist: int = 10,
    seed: int = 0,
    device: str = "cpu",
    regions: Optional[torch.Tensor] = None,
    cluster_size: int = 0,
    cluster_spread: float = 0.05,
    qk_scale: float = 1.0,
    coords_scale: float = 1.0,
) -> Dict[str, torch.Tensor]:
    """One cloud in the calling convention of the reference's src variant.

    The cloud is padded once, at the end, to a multiple of ``block_size``
    (``src/models/baselines/transformer.py:43-57``).  q, k, v of the padding rows are drawn like every other
    row (in the model they are W(norm(0)), not zero): the operator itself has to blank them.
    """
    gen = torch.Generator().manual_seed(seed)
    n = raw_size + (-raw_size) % block_size
    hd = num_heads * h_dim
    q = torch.randn(n, hd, generator=gen)
    k = torch.randn(n, hd, generator=gen)
    v = torch.randn(n, hd, generator=gen)
    coords = torch.randn(raw_size, coords_dim, generator=gen)
    if cluster_size > 0:
        n_clu = (n + cluster_size - 1) // cluster_size
        cluster_id = (torch.randperm(n, generator=gen) // cluster_size).clamp(max=n_clu - 1)
        centres = torch.randn(n_clu, 3 * hd + coords_dim, generator=gen)[cluster_id]
        q = centres[:, :hd] + cluster_spread * q
        k = centres[:, :hd] + cluster_spread * k
        v = centres[:, 2 * hd : 3 * hd] + cluster_spread * v
        coords = centres[:raw_size, 3 * hd :] + cluster_spread * coords
    q, k, coords = q * qk_scale, k * qk_scale, coords * coords_scale
    w_rpe, _ = _linear_init(gen, hd, (coords_dim - 1) * num_w_per_dist, bias=False)
    alpha = torch.randn(num_heads, h_dim + coords_dim, n_hashes, generator=gen)
    out_w, out_b = _linear_init(gen, h_dim, hd, bias=True)
    if regions is None:
        regions = get_regions(num_regions, n_hashes, num_heads, generator=gen)
    _, kw = prepare_input_src(torch.zeros(raw_size, 1), coords, {"block_size": block_size, "regions": regions})
    out = {
        "q": q, "k": k, "v": v, "coords": kw["coords"].contiguous(),
        "eta_idx": kw["region_indices"][0].contiguous(), "phi_idx": kw["region_indices"][1].contiguous(),
        "regions_h": kw["regions_h"].contiguous(), "w_rpe_weight": w_rpe, "alpha": alpha, "out_weight": out_w,
        "out_bias": out_b, "regions": regions, "coords_raw": coords,
    }
    out = {name: t.to(device) for name, t in out.items()}
    out["raw_size"] = raw_size
    return out


# Named workloads = BASELINE.json "configs" (SURVEY.md §8d c1..c5).
WORKLOADS = {
    "example-4k": dict(cloud_sizes=[4096], block_size=64, n_hashes=2, coords_dim=6, num_regions=150),
    "tracking-6k": dict(cloud_sizes=[6000], block_size=128, n_hashes=3, coords_dim=6, num_regions=150),
    # ten tracking-6k clouds in one call: the batch index goes into the AND code (example/transformer.py:55-56)
    "tracking-6k-x10": dict(cloud_sizes=[6000] * 10, block_size=128, n_hashes=3, coords_dim=6, num_regions=150),
    "tracking-60k": dict(cloud_sizes=[60000], block_size=128, n_hashes=3, coords_dim=6, num_regions=150),
    "tracking-60k-t8": dict(cloud_sizes=[60000], block_size=128, n_hashes=8, coords_dim=6, num_regions=150),
    "pileup-8clouds": dict(
        cloud_sizes=[2000, 14000, 5000, 9000, 3000, 12000, 7000, 8000],
        block_size=256, n_hashes=3, coords_dim=4, num_regions=140,
    ),
}


def workload_inputs(name: str, seed: int = 0, device: str = "cpu", **overrides) -> Dict[str, torch.Tensor]:
    cfg = dict(WORKLOADS[name])
    cfg.update(overrides)
    return make_inputs(seed=seed, device=device, **cfg)
