"""Randomised stress of hept_segmented_argsort against torch.sort(stable=True): sizes, segment counts, key
distributions (smooth, quantised, clustered, padded with +-inf, signed zeros).  python tools/sort_stress.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(1234)
dev = torch.device("cuda", 0)
bad = 0
for it in range(iters):
    kind = it % 6
    big = int(torch.randint(0, 10, (1,), generator=g)) == 0
    n = int(torch.randint(1, 200000 if big else 9000, (1,), generator=g))
    s = int(torch.randint(1, 4 if big else 40, (1,), generator=g))
    if kind == 0:
        keys = torch.randn(s, n, generator=g)
    elif kind == 1:
        keys = torch.randint(0, max(2, n // 50), (s, n), generator=g).float()                     # heavy ties
    elif kind == 2:
        keys = torch.randn(s, n, generator=g) * 1e-3 + torch.randint(0, 7, (s, n), generator=g).float() * 100.0
    elif kind == 3:
        keys = torch.randn(s, n, generator=g)
        keys[torch.rand(s, n, generator=g) < 0.2] = float("inf")
        keys[torch.rand(s, n, generator=g) < 0.05] = float("-inf")
    elif kind == 4:
        keys = torch.zeros(s, n)
        keys[torch.rand(s, n, generator=g) < 0.5] = -0.0
        keys[:, : n // 3] += torch.rand(s, n // 3, generator=g) * 1e-30                           # denormal-ish spread
    else:
        keys = torch.exp(torch.randn(s, n, generator=g) * 8.0)                                    # 14 decades
    got = ops.segmented_argsort(keys.to(dev).contiguous()).long().cpu()
    want = torch.sort(keys, dim=-1, stable=True).indices
    if not torch.equal(got, want):
        bad += 1
        print(f"MISMATCH it={it} kind={kind} n={n} s={s}", flush=True)
print(f"{iters} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
