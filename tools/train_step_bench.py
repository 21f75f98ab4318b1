"""fwd + bwd of HEPTAttention (autograd path) at tracking-60k: ms per training step with fp32 tiles (default) and with
the opt-in bf16 tiles (``train_tiles = "bf16"``), and how far the bf16 gradients are from the fp32 ones."""
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


def grads():
    w_rpe.weight.grad = m.out_linear.weight.grad = None   # (step() lets the parameter gradients accumulate)
    out = m(q, k, v, **kw)
    out.backward(gout)
    res = [q.grad.clone(), k.grad.clone(), v.grad.clone(), w_rpe.weight.grad.clone(), m.out_linear.weight.grad.clone()]
    q.grad = k.grad = v.grad = w_rpe.weight.grad = m.out_linear.weight.grad = None
    return out.detach().clone(), res


ref_out, ref = None, None
for tiles in ("fp32", "bf16"):
    m.train_tiles = tiles
    out, gr = grads()
    if tiles == "fp32":
        ref_out, ref = out, gr
    else:   # the 16-bit training mode against the fp32 one: max error relative to each tensor's own scale
        rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip([out] + gr, [ref_out] + ref)]
        print("bf16 tiles vs fp32 tiles, max error / tensor scale: out %.2e  dq %.2e  dk %.2e  dv %.2e  dw_rpe %.2e  dW_out %.2e" % tuple(rel))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print(f"train step (fwd+bwd, {tiles} tiles): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
