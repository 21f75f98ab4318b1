"""fwd + bwd of HEPTAttention (fp32 tiles, autograd path) at tracking-60k: ms per training step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10).to(dev).train()
with torch.no_grad():
    m.e2lsh.alpha.copy_(g["alpha"])
w_rpe = torch.nn.Linear(50, 192).to(dev)
q, k, v = (g[x].clone().requires_grad_(True) for x in ("q", "k", "v"))
kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
gout = torch.randn(q.shape[0], 24, device=dev)


def step():
    out = m(q, k, v, **kw)
    out.backward(gout)
    q.grad = k.grad = v.grad = None


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print(f"train step (fwd+bwd, fp32 tiles): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
