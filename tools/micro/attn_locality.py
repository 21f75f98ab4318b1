"""How much of the block attention's time is the RANDOMNESS of its gathers: the same launch (tracking-60k rows, bf16 and
f32 tiles) with the forward's own permutations, with identity permutations (every gather and scatter sequential) and with
uniformly random ones.  python tools/micro/attn_locality.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hept_amd import ops  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
n, h, t = g["q"].shape[0], 8, 3
sq = ops.rpe_scale(g["w_rpe_weight"], h, 24, 10)
for prec in ("bf16", "fp32"):
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sq, g["alpha"], g["combined_shifts"], prec)
    qp, kp = ops.sort_tables(ph["qproj"], ph["kproj"], g["combined_shifts"], ph["minmax"])
    ident = torch.arange(n, device=dev, dtype=torch.int32).expand(t, h, n).contiguous()
    gen = torch.Generator(device="cpu").manual_seed(1)
    rnd = torch.stack([torch.randperm(n, generator=gen) for _ in range(t * h)]).view(t, h, n).to(dev).to(torch.int32)
    rnd2 = torch.stack([torch.randperm(n, generator=gen) for _ in range(t * h)]).view(t, h, n).to(dev).to(torch.int32)
    for name, a, b in (("forward's own", qp, kp), ("identity", ident, ident), ("random", rnd, rnd2),
                       ("own q, identity k", qp, ident), ("identity q, own k", ident, kp)):
        for _ in range(5):
            ops.block_attn(ph["qhat"], ph["kvhat"], a, b, 24, 128)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.block_attn(ph["qhat"], ph["kvhat"], a, b, 24, 128)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        print(f"{prec} {name:20s}: {best:7.1f} us per launch (ops.block_attn incl. its output allocation)", flush=True)
