"""Attn block at tracking-60k on one GPU: fused (one C call) vs the reference composition (torch LayerNorm /
Linear around HEPTAttention).  python tools/attn_block_bench.py [bf16|fp32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import Attn  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
n = inp["q"].shape[0]
torch.manual_seed(0)
blk = Attn(6, precision=prec, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10).to(dev).eval()
with torch.no_grad():
    blk.attn.e2lsh.alpha.copy_(inp["alpha"])
    blk.w_q.weight.mul_(0.3)
    blk.w_k.weight.mul_(0.3)
x = torch.randn(n, 24, device=dev)
kwargs = {"coords": inp["coords"].to(dev), "combined_shifts": inp["combined_shifts"].to(dev)}


def composed(x):
    x_normed = blk.norm1(x)
    q, k, v = blk.w_q(x_normed), blk.w_k(x_normed), blk.w_v(x_normed)
    aggr = blk.attn(q, k, v, pe=kwargs["coords"], w_rpe=blk.w_rpe, **kwargs)
    x = x + aggr
    return x + blk.ff(blk.norm2(x))


def timeit(fn, reps=200):
    with torch.no_grad():
        for _ in range(20):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


with torch.no_grad():
    a, b = blk(x, kwargs), composed(x)
err = (a - b).abs().amax(-1)
print(f"fused vs composed: rows within 1e-3: {(err <= 1e-3 * (b.abs().amax(-1) + 1)).float().mean():.4f}  max {err.max():.3e}")
print(f"{prec}: fused block {timeit(lambda t: blk(t, kwargs)):.1f} us   composed {timeit(composed):.1f} us   "
      f"(N_raw {inp['n_raw']}, padded {n})")
