"""Per-stage time of one forward (HIP events inside hept_forward) for a named workload: python tools/stage_times.py
[workload ...] [--precision bf16]."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention, ops  # noqa: E402
from hept_amd.synthetic import WORKLOADS, workload_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=list(WORKLOADS))
ap.add_argument("--precision", default="bf16")
args = ap.parse_args()
dev = torch.device("cuda", 0)
for name in args.workloads:
    inp = workload_inputs(name, seed=0)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    h, e, t = inp["alpha"].shape
    c = inp["coords"].shape[1]
    w_rpe = torch.nn.Linear((c - 1) * 10, 192).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(g["w_rpe_weight"])
    m = HEPTAttention(e, h_dim=24, num_heads=h, block_size=WORKLOADS[name]["block_size"], n_hashes=t, num_w_per_dist=10,
                      precision=args.precision)
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
    m = m.to(dev).eval()
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        for _ in range(5):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        ops.profile_enable(2, 20)
        for _ in range(20):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        ms, n = ops.profile_read()
        ops.profile_enable(0)
    print(f"{name:16s} {args.precision}: " + "  ".join(f"{k} {v / n * 1e3:7.1f} us" for k, v in ms.items()), flush=True)
