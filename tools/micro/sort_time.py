"""us per hept_sort_tables call (both kernels, back to back on one stream, HIP events) + an exactness check of the
result against torch.sort(stable=True) of the same keys.  python tools/micro/sort_time.py [workload] [label]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hept_amd import ops
from hept_amd.synthetic import workload_inputs

wl = sys.argv[1] if len(sys.argv) > 1 else "tracking-60k"
label = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
inp = workload_inputs(wl, seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
r = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision="bf16")
qpos, kpos = ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
mm = r["minmax"]
span = mm[..., 1].amax(-1) - mm[..., 0].amin(-1)
offs = g["combined_shifts"].float() * span[..., None]
ok = all(torch.equal(pos.long(), torch.sort(proj + offs, dim=-1, stable=True).indices)
         for pos, proj in ((qpos, r["qproj"]), (kpos, r["kproj"])))
best = 1e9
for rep in range(3):
    for _ in range(20):
        ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
print(f"{label or 'default'} {wl}: {best:.1f} us per sort, exact={ok}", flush=True)
