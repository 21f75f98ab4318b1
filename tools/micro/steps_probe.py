import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hept_amd import HEPTAttention
from hept_amd.synthetic import workload_inputs
dev = torch.device("cuda:0")
def build(T):
    inp = workload_inputs("tracking-60k", seed=0, n_hashes=T)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    attn = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=T, num_w_per_dist=10, precision="bf16")
    attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
    attn = attn.to(dev).eval()
    attn.reserve(inp["q"].shape[0], 6, dev)
    w = torch.nn.Linear(50, 192).to(dev)
    with torch.no_grad(): w.weight.copy_(g["w_rpe_weight"])
    kw = dict(w_rpe=w, coords=g["coords"], combined_shifts=g["combined_shifts"])
    def step():
        with torch.no_grad(): return attn(g["q"], g["k"], g["v"], **kw)
    return step
s3 = build(3); s1 = build(1)
for _ in range(300): s3()
torch.cuda.synchronize()
for trial in range(3):
    for _ in range(200): s1()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(27)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(26):
        s3(); evs[i + 1].record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("trial", trial, "per-step us:", " ".join(f"{evs[i].elapsed_time(evs[i+1])*1e3:.0f}" for i in range(26)), f"| wall {(t1-t0)/26*1e6:.1f}")
# the contract's protocol: W=5, K=20, wall clock between synchronizes
for trial in range(3):
    for _ in range(200): s1()
    torch.cuda.synchronize()
    for _ in range(5): s3()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): s3()
    torch.cuda.synchronize()
    print("protocol W=5 K=20:", f"{(time.perf_counter()-t0)/20*1e6:.1f} us/step")
for trial in range(2):
    for _ in range(200): s1()
    torch.cuda.synchronize()
    for _ in range(5): s3()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): s3()
    torch.cuda.synchronize()
    print("protocol W=5 K=500:", f"{(time.perf_counter()-t0)/500*1e6:.1f} us/step")
