"""What the forward would cost if its inputs started in (pinned) host memory: H2D copy of q, k, v, coords, codes +
forward.  Reported in DESIGN.md §6 for reference; never part of bench.py's `value` (the boundary takes device pointers)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
host = {k: inp[k].contiguous().pin_memory() for k in ("q", "k", "v", "coords", "combined_shifts")}
nbytes = sum(t.numel() * t.element_size() for t in host.values())
m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16").to(dev).eval()
w_rpe = torch.nn.Linear(50, 192).to(dev)
devbuf = {k: torch.empty_like(v, device=dev) for k, v in host.items()}


def step():
    for k_, v_ in host.items():
        devbuf[k_].copy_(v_, non_blocking=True)
    return m(devbuf["q"], devbuf["k"], devbuf["v"], w_rpe=w_rpe, coords=devbuf["coords"],
             combined_shifts=devbuf["combined_shifts"])


with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print(f"host-resident inputs ({nbytes / 1e6:.0f} MB H2D per forward): {dt * 1e3:.3f} ms/forward, "
      f"{inp['n_raw'] / dt / 1e6:.1f} M points/s, H2D {nbytes / dt / 1e9:.1f} GB/s effective")
