"""A dozen forwards of one named workload (for the PMC passes of the block-attention kernel, one per record of bench.py's
line: tools/pmc_all.sh).  python tools/attn_shape_run.py <workload> <precision> [block size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hept_amd import HEPTAttention
from hept_amd.synthetic import WORKLOADS, workload_inputs

wl, prec = sys.argv[1], sys.argv[2]
kw = {"block_size": int(sys.argv[3])} if len(sys.argv) > 3 else {}
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
    for _ in range(12):
        attn(g["q"], g["k"], g["v"], w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
torch.cuda.synchronize()
