import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import workload_inputs, WORKLOADS
dev = torch.device("cuda", 0)
def build(wl, prec):
    inp = workload_inputs(wl, seed=0)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    c = inp["coords"].shape[1]
    m = HEPTAttention(24 + c, h_dim=24, num_heads=8, block_size=WORKLOADS[wl]["block_size"], n_hashes=inp["alpha"].shape[2], num_w_per_dist=10, precision=prec)
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]}, strict=True)
    m = m.to(dev).eval()
    w = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad(): w.weight.copy_(g["w_rpe_weight"])
    kw = dict(w_rpe=w, coords=g["coords"], combined_shifts=g["combined_shifts"])
    def step():
        with torch.no_grad(): return m(g["q"], g["k"], g["v"], **kw)
    return step
for wl in ("example-4k", "tracking-6k"):
    for prec in ("fp32", "bf16", "bf16", "fp32"):
        step = build(wl, prec)
        for _ in range(10): step()
        torch.cuda.synchronize()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(200): step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        print(wl, prec, " ".join(f"{t:.1f}" for t in ts), flush=True)
