"""us per hept_combine_out call at tracking-60k (3 tables, packed or f32 rows; HIP events, back-to-back calls) and a
bit-exactness digest of the output.  python tools/micro/combine_time.py [bf16|fp32] [label]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hept_amd import ops
from hept_amd.synthetic import workload_inputs

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
label = sys.argv[2] if len(sys.argv) > 2 else "default"
dev = torch.device("cuda:0")
inp = workload_inputs("tracking-60k", seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
r = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision=prec)
qpos, kpos = ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
part = ops.block_attn(r["qhat"], r["kvhat"], qpos, kpos, 24, 128)
out = ops.combine_out(part, 24, g["out_weight"], g["out_bias"])
digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
best = 1e9
for rep in range(5):
    for _ in range(20):
        ops.combine_out(part, 24, g["out_weight"], g["out_bias"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.combine_out(part, 24, g["out_weight"], g["out_bias"])
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
print(f"{label} {prec}: {best:.1f} us per combine_out, out sha {digest}", flush=True)
