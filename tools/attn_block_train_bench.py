"""Training step (forward + backward) of ONE Attn block at tracking-60k on one GPU: norm1 / w_q / w_k / w_v folded into
the operator's row builder (one autograd node, ``Attn.fuse_training``) against the reference's composition of torch
modules around the operator.  python tools/attn_block_train_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import Attn  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
n = inp["q"].shape[0]
torch.manual_seed(0)
blk = Attn(6, precision="fp32", h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10).to(dev).train()
with torch.no_grad():
    blk.attn.e2lsh.alpha.copy_(inp["alpha"])
    blk.w_q.weight.mul_(0.3)
    blk.w_k.weight.mul_(0.3)
x = torch.randn(n, 24, device=dev, requires_grad=True)
gout = torch.randn(n, 24, device=dev)
kwargs = {"coords": inp["coords"].to(dev), "combined_shifts": inp["combined_shifts"].to(dev)}


def step():
    y = blk(x, kwargs)
    y.backward(gout)
    x.grad = None
    for p in blk.parameters():
        p.grad = None


for fuse, tiles in ((True, "fp32"), (False, "fp32"), (True, "fp32"), (False, "fp32"), (True, "bf16")):
    blk.fuse_training = fuse
    blk.attn.train_tiles = tiles   # "bf16": the opt-in 16-bit kernels in both directions
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print(f"Attn block train step (fwd+bwd, {tiles} tiles, dropout 0.1), norm1 + projections "
          f"{'folded into the row builder' if fuse else 'composed (torch modules)'}: "
          f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms", flush=True)
