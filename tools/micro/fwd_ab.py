"""us per HEPTAttention.forward (whole operator, in place: the gather kernels run cold, as in bench.py) + the stage
times, for A/B builds of one kernel.  The first call for a (workload, precision) saves the output; later calls print
the largest difference to it (absolute, and relative to atol 1e-5 + rtol 1e-4: the fp32 every-element bound).
python tools/micro/fwd_ab.py [precision] [label] [workload] [block size | -] [n_hashes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import workload_inputs, WORKLOADS

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
label = sys.argv[2] if len(sys.argv) > 2 else "default"
wl = sys.argv[3] if len(sys.argv) > 3 else "tracking-60k"
kw = {}
if len(sys.argv) > 4 and sys.argv[4] not in ("-", ""):
    kw["block_size"] = int(sys.argv[4])
if len(sys.argv) > 5:
    kw["n_hashes"] = int(sys.argv[5])
dev = torch.device("cuda:0")
inp = workload_inputs(wl, seed=0, **kw)
bs = kw.get("block_size", WORKLOADS[wl]["block_size"])
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
c = inp["coords"].shape[1]
attn = HEPTAttention(24 + c, h_dim=24, num_heads=8, block_size=bs, n_hashes=inp["alpha"].shape[2], num_w_per_dist=10,
                     precision=prec)
attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                      "e2lsh.alpha": inp["alpha"]}, strict=True)
attn = attn.to(dev).eval()
w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
with torch.no_grad():
    w_rpe.weight.copy_(g["w_rpe_weight"])
args = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])


def step():
    with torch.no_grad():
        return attn(g["q"], g["k"], g["v"], **args)


out = step()
torch.cuda.synchronize()
base = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out",
                    f"fwd_ab_base_{wl}_{bs}_{inp['alpha'].shape[2]}_{prec}.pt")
diff = ""
if os.path.exists(base):
    ref = torch.load(base).to(dev)
    d = (out - ref).abs()
    tol = 1e-5 + 1e-4 * ref.abs()
    diff = f", vs base: max abs {d.max().item():.3e}, max err/tol {(d / tol).max().item():.3f}, finite {bool(torch.isfinite(out).all())}"
else:
    os.makedirs(os.path.dirname(base), exist_ok=True)
    torch.save(out.cpu(), base)
best = 1e9
for rep in range(3):
    for _ in range(20):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        step()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
ops.profile_enable(2, 64)
for _ in range(50):
    step()
torch.cuda.synchronize()
ms, cnt = ops.profile_read()
ops.profile_enable(0, 0)
st = " ".join(f"{k.replace('_tables','').replace('block_','')}={v / cnt * 1e3:.1f}" for k, v in ms.items() if v > 0)
print(f"{label} {wl} B={bs} T={inp['alpha'].shape[2]} {prec}: {best:.1f} us per forward [{st}]{diff}", flush=True)
