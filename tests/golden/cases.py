"""Golden-vector cases for the HEPT hot path.

Shared by ``make_golden.py`` (which runs the real reference on these inputs in
the build container) and by the tests (which rebuild the same inputs from the
seeds plus the small arrays stored in each ``*.npz``).  Nothing here touches
``/root/reference``.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from hept_amd.synthetic import make_inputs, make_inputs_src  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# name -> static description.  "stored" keys are replayed from the fixture.
CASES: Dict[str, dict] = {
    # G1: tiny, default-init weights, random AND codes in [0, 16); everything stored.
    "g1_rand512": dict(
        cloud_sizes=[512], block_size=64, n_hashes=2, coords_dim=6, num_regions=150, seed=101,
        random_codes=16,
    ),
    # G2: BASELINE config 1 shape (N=4096, B=64, T=2); 4 imbalanced clouds, codes from prepare_input.
    "g2_example4k": dict(
        cloud_sizes=[700, 1500, 1200, 600], block_size=64, n_hashes=2, coords_dim=6, num_regions=150, seed=102,
    ),
    # G3: tracking-6k shape with the shipped checkpoint's layer-0 weights (trained-weight numerics).
    "g3_ckpt6k": dict(
        cloud_sizes=[6000], block_size=128, n_hashes=3, coords_dim=6, num_regions=150, seed=103,
        ckpt_weights=True, cluster_size=8, qk_scale=0.08, coords_gain=10.0,
    ),
    # G7: G3's cloud and checkpoint weights with the coordinates LEFT ALONE (N(0,1) per column): the shipped layer-0
    # scales (sqrt_w up to 5.8e3) then put |q^|^2 at ~1e8, almost every weight underflows and the surviving logits are
    # differences of 1e8-sized terms -- a finite-output / no-overflow check of every precision, not a tight parity case.
    "g7_ckpt_rawcoords": dict(
        cloud_sizes=[6000], block_size=128, n_hashes=3, coords_dim=6, num_regions=150, seed=103,
        ckpt_weights=True, cluster_size=8, qk_scale=0.08, no_grads=True,
    ),
    # G4: pileup shape (C=4, E=28, B=256), imbalanced clouds each >= B, batch index in the AND code.
    "g4_pileup": dict(
        cloud_sizes=[300, 900, 520, 1400, 260, 700], block_size=256, n_hashes=3, coords_dim=4,
        num_regions=140, seed=104, qk_scale=0.3, coords_scale=0.05,
    ),
    # G5: tracking-60k shape (BASELINE config 3); only sampled output rows are stored.
    "g5_track60k": dict(
        cloud_sizes=[60000], block_size=128, n_hashes=3, coords_dim=6, num_regions=150, seed=105,
        sample_rows=1024,
    ),
    # G6: block_size 100 as in the reference's own yaml (not a multiple of the 32-row MFMA tile).
    "g6_block100": dict(
        cloud_sizes=[950, 1333], block_size=100, n_hashes=3, coords_dim=6, num_regions=150, seed=106,
        qk_scale=0.25, coords_scale=0.1,
    ),
}

NUM_HEADS, H_DIM, W_PER_DIST = 8, 24, 10

# Cases of the reference's src variant (src/models/attention/hept.py with the caller-side
# preparation of src/models/baselines/transformer.py:43-57): one cloud, padded at the end.
SRC_CASES: Dict[str, dict] = {
    # S1: small, 24 padding rows, everything stored
    "s1_src1000": dict(raw_size=1000, block_size=128, n_hashes=3, coords_dim=6, num_regions=150, seed=201,
                       qk_scale=0.3, coords_scale=0.1),
    # S2: block size 100 of the reference's own yaml; raw size a multiple of it (no padding rows)
    "s2_src5000": dict(raw_size=5000, block_size=100, n_hashes=3, coords_dim=6, num_regions=150, seed=202,
                       cluster_size=8, qk_scale=0.25, coords_scale=0.3),
    # S3: pileup shape (C=4, B=256), 187 padding rows
    "s3_src_pileup": dict(raw_size=2885, block_size=256, n_hashes=3, coords_dim=4, num_regions=140, seed=203,
                          qk_scale=0.3, coords_scale=0.05),
}


def build_inputs_src(name: str, stored: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, torch.Tensor]:
    """Inputs of src-variant case ``name``; ``stored`` supplies ``regions`` (the reference's own draw)."""
    cfg = dict(SRC_CASES[name])
    stored = stored or {}
    regions = torch.from_numpy(np.asarray(stored["regions"])).float() if "regions" in stored else None
    raw_size = cfg.pop("raw_size")
    inp = make_inputs_src(raw_size, num_heads=NUM_HEADS, h_dim=H_DIM, num_w_per_dist=W_PER_DIST, regions=regions,
                          **cfg)
    for key in ("eta_patch", "phi_patch"):  # tie-induced differences from the reference's unstable argsort
        if key + "_idx" in stored and len(stored[key + "_idx"]):
            idx = torch.from_numpy(np.asarray(stored[key + "_idx"]).astype(np.int64))
            inp[key[:3] + "_idx"][tuple(idx.T)] = torch.from_numpy(np.asarray(stored[key + "_val"])).float()
    inp["block_size"] = cfg["block_size"]
    inp["w_per_dist"] = W_PER_DIST
    return inp


def load_case_src(name: str):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    return build_inputs_src(name, fx), fx


def build_inputs(name: str, stored: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, torch.Tensor]:
    """Inputs of case ``name``.  ``stored`` supplies the replayed arrays
    (``regions``, ``pad_seq``, and for G3 the checkpoint weights)."""
    cfg = CASES[name]
    stored = stored or {}
    regions = torch.from_numpy(np.asarray(stored["regions"])).float() if "regions" in stored else None
    pad_seq = torch.from_numpy(np.asarray(stored["pad_seq"]).astype(np.int64)) if "pad_seq" in stored else None
    inp = make_inputs(
        cfg["cloud_sizes"], block_size=cfg["block_size"], n_hashes=cfg["n_hashes"],
        coords_dim=cfg["coords_dim"], num_heads=NUM_HEADS, h_dim=H_DIM, num_regions=cfg["num_regions"],
        num_w_per_dist=W_PER_DIST, seed=cfg["seed"], regions=regions, pad_seq=pad_seq,
        cluster_size=cfg.get("cluster_size", 0),
    )
    gen = torch.Generator().manual_seed(cfg["seed"] + 7919)
    if cfg.get("random_codes"):
        t, h, n = inp["combined_shifts"].shape
        inp["combined_shifts"] = torch.randint(0, cfg["random_codes"], (t, h, n), generator=gen)
    if cfg.get("ckpt_weights"):
        # trained layer-0 weights replayed from the fixture; q,k,v = W(LayerNorm(x)) as in the model shell
        for key in ("w_rpe_weight", "alpha", "out_weight", "out_bias"):
            inp[key] = torch.from_numpy(np.asarray(stored[key])).float()
        n_raw = inp["n_raw"]
        n_clu = int(inp["cluster_id"].max()) + 1
        x = torch.randn(n_clu, H_DIM, generator=gen)[inp["cluster_id"]] + 0.05 * torch.randn(n_raw, H_DIM, generator=gen)
        xn = torch.nn.functional.layer_norm(
            x, (H_DIM,), torch.from_numpy(np.asarray(stored["norm1_weight"])).float(),
            torch.from_numpy(np.asarray(stored["norm1_bias"])).float(),
        )
        ps = inp["pad_seq"]
        inp["x"] = x[ps].contiguous()  # block-level input (Attn cases)
        for key, wkey in (("q", "w_q"), ("k", "w_k"), ("v", "w_v")):
            w = torch.from_numpy(np.asarray(stored[wkey])).float()
            inp[key] = torch.nn.functional.linear(xn, w)[ps].contiguous()
        # trained coordinate weights span 1e-3 .. 6e3 per (head, coordinate): bring every
        # coordinate to O(coords_gain) for the head that weighs it most (real detector features
        # are normalised per column; the dataset itself is not available here).  |q̂|² then
        # reaches ~1e3, which exercises the cancellation in q·k - ½|q|² - ½|k|².
        if cfg.get("coords_gain"):
            w4 = inp["w_rpe_weight"].reshape(NUM_HEADS, H_DIM, -1, W_PER_DIST)
            qw = w4.sum(1).clamp(max=50).exp().sum(-1)
            sqrt_w = torch.sqrt(2 * torch.cat([qw[:, :1], qw], dim=-1))
            per_dim = cfg["coords_gain"] / sqrt_w.max(dim=0).values
            inp["coords"] = (inp["coords"] * per_dim).contiguous()
            inp["coords_raw"] = inp["coords_raw"] * per_dim
    if "code_patch_idx" in stored and len(stored["code_patch_idx"]):
        idx = torch.from_numpy(np.asarray(stored["code_patch_idx"]).astype(np.int64))
        inp["combined_shifts"][tuple(idx.T)] = torch.from_numpy(np.asarray(stored["code_patch_val"]))
    if cfg.get("qk_scale"):
        # well-conditioned variant: neighbours inside a block carry O(1) weight
        inp["q"] = inp["q"] * cfg["qk_scale"]
        inp["k"] = inp["k"] * cfg["qk_scale"]
    if cfg.get("coords_scale"):
        inp["coords"] = (inp["coords"] * cfg["coords_scale"]).contiguous()
        inp["coords_raw"] = inp["coords_raw"] * cfg["coords_scale"]
    inp["block_size"] = cfg["block_size"]
    inp["w_per_dist"] = W_PER_DIST
    return inp


# Attn-block cases (reference example/transformer.py:131-165, eval mode): the block input x (N, 24),
# coords and AND codes of an operator case, and a full set of block weights stored in the fixture.
ATTN_CASES: Dict[str, dict] = {
    # A1: the shipped checkpoint's layer 0 (attns.0.*, w_q and w_k scaled by G3's qk_scale: see make_golden_attn.py)
    # on the tracking-6k cloud of G3
    "a1_attn_ckpt6k": dict(base="g3_ckpt6k"),
    # A2: default-initialised block, block_size 100, two clouds (G6's coords and codes), x ~ N(0,1)
    "a2_attn_rand": dict(base="g6_block100", seed=301),
}
ATTN_KEYS = ("norm1.weight", "norm1.bias", "w_q.weight", "w_k.weight", "w_v.weight", "w_rpe.weight", "w_rpe.bias",
             "attn.e2lsh.alpha", "attn.out_linear.weight", "attn.out_linear.bias", "norm2.weight", "norm2.bias",
             "ff.0.weight", "ff.0.bias", "ff.2.weight", "ff.2.bias")


def build_inputs_attn(name: str, stored: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, torch.Tensor]:
    """x, coords, combined_shifts (+ ``params`` when the fixture is given) of Attn-block case ``name``."""
    cfg = ATTN_CASES[name]
    base, _ = load_case(cfg["base"])
    out = {"coords": base["coords"], "combined_shifts": base["combined_shifts"], "block_size": base["block_size"],
           "w_per_dist": W_PER_DIST}
    if "x" in base:
        out["x"] = base["x"]
    else:
        gen = torch.Generator().manual_seed(cfg["seed"])
        out["x"] = torch.randn(base["q"].shape[0], H_DIM, generator=gen)
    if stored is not None:
        out["params"] = {k: torch.from_numpy(np.asarray(stored["p:" + k])).float() for k in ATTN_KEYS}
    return out


def load_case_attn(name: str):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    return build_inputs_attn(name, fx), fx


def input_checksums(inp: Dict[str, torch.Tensor]) -> np.ndarray:
    """Order-sensitive float64 checksums of the generated inputs (detects RNG drift)."""
    vals = []
    for key in ("q", "k", "v", "coords", "w_rpe_weight", "alpha", "out_weight", "out_bias"):
        t = inp[key].double().flatten()
        w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 8191
        vals.append(float((t * w).sum()))
    if "combined_shifts" in inp:
        vals.append(float(inp["combined_shifts"].double().sum()))
    else:
        vals.append(float(inp["eta_idx"].double().sum() + 3 * inp["phi_idx"].double().sum()))
    return np.asarray(vals, dtype=np.float64)


def load_case(name: str):
    """(inputs, fixture dict) for a committed golden case."""
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    return build_inputs(name, fx), fx
