"""Step time of a freshly built module, step by step, on a device another module has kept busy (why a 5-step warm-up
is short of the steady state).  python tools/fresh_module_steps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import workload_inputs

dev = torch.device("cuda:0")
inp = workload_inputs("tracking-60k", seed=0, n_hashes=3)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
c = inp["coords"].shape[1]

def make():
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
    return step

a = make()
for _ in range(500):
    a()
torch.cuda.synchronize()
for trial in range(2):
    b = make()
    for _ in range(200):
        a()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
    for i in range(80):
        ev[i].record()
        b()
    ev[80].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(80)]
    print("fresh module, us per step:", " ".join(f"{x:.0f}" for x in d))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    for i in range(40):
        ev[i].record()
        a()
    ev[40].record()
    torch.cuda.synchronize()
    print("old module again:         ", " ".join(f"{ev[i].elapsed_time(ev[i + 1]) * 1e3:.0f}" for i in range(40)))
