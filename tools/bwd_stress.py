"""Randomised shapes of the backward: the split-bf16 kernel (1e-4) and the 16-bit-tile kernel (0.1 of each tensor's
scale) against the native f32 MFMA kernel (all HIP), block
sizes that are not multiples of 32, 1..6 tables, every supported (head_dim, coords_dim) pair, one or several clouds.
python tools/bwd_stress.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hept_amd import ops  # noqa: E402
from hept_amd.synthetic import make_inputs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(7)
dev = torch.device("cuda", 0)
pairs = [(24, 6), (24, 4), (24, 2), (16, 6), (16, 4), (8, 4)]
bad = 0
for it in range(iters):
    d, c = pairs[it % len(pairs)]
    b = int(torch.randint(8, 257, (1,), generator=g))
    t = int(torch.randint(1, 7, (1,), generator=g))
    n_clouds = int(torch.randint(1, 4, (1,), generator=g))
    sizes = [int(torch.randint(b, 4 * b + 40, (1,), generator=g)) for _ in range(n_clouds)]
    inp = make_inputs(sizes, block_size=b, n_hashes=t, coords_dim=c, h_dim=d, seed=2000 + it)
    gd = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    gd["q"], gd["k"], gd["coords"] = gd["q"] * 0.3, gd["k"] * 0.3, gd["coords"] * 0.2
    h = gd["alpha"].shape[0]
    sw = ops.rpe_scale(gd["w_rpe_weight"], h, d, 10)
    ph = ops.prep_hash(gd["q"], gd["k"], gd["v"], gd["coords"], sw, gd["alpha"], gd["combined_shifts"], "fp32")
    qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], gd["combined_shifts"], ph["minmax"])
    n = gd["q"].shape[0]
    gacc = torch.zeros(n, h, 32, device=dev)
    gacc[..., :d + 1] = torch.randn(n, h, d + 1, generator=torch.Generator().manual_seed(it)).to(dev)
    got = ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, d, c, b)
    ref = ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, d, c, b, f32_mfma=True)
    worst = 0.0
    for a, r in zip(got, ref):
        worst = max(worst, float((a - r).abs().max()) / (float(r.abs().max()) + 1e-30))
        if not bool(torch.isfinite(a).all()):
            worst = float("inf")
    if not worst <= 1e-4:
        bad += 1
        print(f"MISMATCH it={it} D={d} C={c} B={b} T={t} sizes={sizes}: worst rel {worst:.3e}", flush=True)
    # the 16-bit training tiles (block_attn_bwd_bf16_kernel on the rows of the bf16 row builder) against the same f32
    # reference: bf16-level agreement on every tensor, finite everywhere
    ph16 = ops.prep_hash(gd["q"], gd["k"], gd["v"], gd["coords"], sw, gd["alpha"], gd["combined_shifts"], "bf16")
    got16 = ops.block_attn_bwd(ph16["qhat"], ph16["kvhat"], qpos, kpos, gacc, d, c, b)
    worst16 = 0.0
    for a, r in zip(got16, ref):
        worst16 = max(worst16, float((a - r).abs().max()) / (float(r.abs().max()) + 1e-30))
        if not bool(torch.isfinite(a).all()):
            worst16 = float("inf")
    if not worst16 <= 0.1:
        bad += 1
        print(f"MISMATCH (bf16 tiles) it={it} D={d} C={c} B={b} T={t} sizes={sizes}: worst rel {worst16:.3e}", flush=True)
print(f"{iters} shapes, {bad} mismatches")
sys.exit(1 if bad else 0)
