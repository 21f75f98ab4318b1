"""Caller-side preparation on the GPU (hept_prepare_input): time per call for the tracking-60k cloud
and the 8-cloud pileup batch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import get_regions, prepare_input  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for name, sizes, b, c, regions_n in (("tracking-60k", [60000], 128, 6, 150),
                                     ("pileup-8clouds", [2000, 14000, 5000, 9000, 3000, 12000, 7000, 8000], 256, 4, 140)):
    n = sum(sizes)
    coords = torch.randn(n, c, generator=g).to(dev)
    x = torch.randn(n, 24, generator=g).to(dev)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(dev)
    helper = {"block_size": b, "num_heads": 8, "regions": get_regions(regions_n, 3, 8, generator=g).to(dev)}
    for _ in range(5):
        prepare_input(x, coords, batch, helper)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        prepare_input(x, coords, batch, helper)
    torch.cuda.synchronize()
    print(f"{name}: prepare_input {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call ({n} points, {len(sizes)} clouds)")
