"""Fixed cost of a timed region: total time of K back-to-back forwards for several K, least-squares a + b*K."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention
from hept_amd.synthetic import workload_inputs
dev = torch.device("cuda", 0)
inp = workload_inputs("tracking-60k", seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16")
m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
m = m.to(dev).eval()
w_rpe = torch.nn.Linear(50, 192).to(dev)
kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
def run(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(k):
            m(g["q"], g["k"], g["v"], **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, t1 - t0
run(50)
ks = [25, 50, 100, 200, 400, 800]
res = {k: min(run(k) for _ in range(5)) for k in ks}
for k in ks:
    print(f"K={k:4d}: total {res[k][0]*1e3:8.3f} ms  ({res[k][0]/k*1e6:6.1f} us/step)   host issue {res[k][1]*1e3:8.3f} ms", flush=True)
a = np.polyfit(ks, [res[k][0] for k in ks], 1)
print(f"fit: {a[0]*1e6:.1f} us/step + {a[1]*1e6:.0f} us fixed")
