"""First forward of a freshly built module on a busy device: host time and GPU time (events), with the workspace
untouched / touched beforehand.  python tools/first_call.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import workload_inputs
dev = torch.device("cuda:0")
inp = workload_inputs("tracking-60k", seed=0, n_hashes=3)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
c = inp["coords"].shape[1]
def make(touch):
    attn = HEPTAttention(24 + c, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16")
    attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
    attn = attn.to(dev).eval()
    attn.reserve(inp["q"].shape[0], c, dev)
    if touch:
        attn._workspace.zero_()
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(g["w_rpe_weight"])
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    def step():
        with torch.no_grad():
            return attn(g["q"], g["k"], g["v"], **kw)
    return step, attn
a, _ = make(False)
for _ in range(50): a()
torch.cuda.synchronize()
keep = []
for touch in (False, True, False, True):
    b, mod = make(touch)
    keep.append(mod)
    torch.cuda.synchronize()
    print(f"workspace {mod._workspace.numel() / 1e6:.0f} MB, touched beforehand: {touch}")
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); b(); e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"  call {i}: host {(t1 - t0) * 1e3:.2f} ms, GPU {e0.elapsed_time(e1):.2f} ms")
