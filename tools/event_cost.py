"""What an event bracket around block_attn costs the step (bench.py samples a stride of its timed steps): the forward timed
with every `stride`-th step bracketed, and the host's enqueue time per step.  python tools/event_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import workload_inputs

dev = torch.device("cuda:0")
inp = workload_inputs("tracking-60k", seed=0, n_hashes=3)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
c = inp["coords"].shape[1]
attn = HEPTAttention(24 + c, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16")
attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
attn = attn.to(dev).eval()
w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
with torch.no_grad():
    w_rpe.weight.copy_(g["w_rpe_weight"])
kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
def step():
    with torch.no_grad():
        return attn(g["q"], g["k"], g["v"], **kw)
for _ in range(200):
    step()
torch.cuda.synchronize()
K = 400
for stride in (0, 1, 2, 5, 20):
    if stride:
        ops.profile_enable(1, K, stride=stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if stride:
        ms, n = ops.profile_read()
        ops.profile_enable(0)
        extra = f"  block_attn by events {ms['block_attn'] / n * 1e3:.1f} us ({n} samples)"
    else:
        extra = ""
    print(f"stride {stride:2d}: {(t2 - t0) / K * 1e6:.1f} us/step, host enqueue {(t1 - t0) / K * 1e6:.1f} us/step{extra}")

# fixed cost of a short timed region (synchronize, K steps, synchronize): the intercept of time against K
import statistics
for K in (1, 2, 5, 10, 20, 50, 200):
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"region of {K:3d} steps: median {statistics.median(ts) * 1e6:.1f} us = {statistics.median(ts) / K * 1e6:.1f} us/step")

# a GPU that idled before the W warm-up steps: how many steps until the step time is back
for idle_ms in (0, 1, 5, 50, 500):
    for W in (5, 50):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            time.sleep(idle_ms * 1e-3)
            for _ in range(W):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"idle {idle_ms:3d} ms, {W:2d} warm-up steps, region of 20: median {statistics.median(ts) / 20 * 1e6:.1f} us/step "
              f"(min {min(ts) / 20 * 1e6:.1f}, max {max(ts) / 20 * 1e6:.1f})")
