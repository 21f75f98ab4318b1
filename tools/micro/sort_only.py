"""prep_hash + sort_tables only, 60 times (for a rocprofv3 kernel trace of the sort stage by itself).
python tools/micro/sort_only.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hept_amd import ops
from hept_amd.synthetic import workload_inputs
wl = sys.argv[1] if len(sys.argv) > 1 else "tracking-6k"
dev = torch.device("cuda:0")
inp = workload_inputs(wl, seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
r = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision="bf16")
for _ in range(60):
    ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
torch.cuda.synchronize()
