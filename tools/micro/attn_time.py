"""us per hept_block_attn call at 60 000 points, 3 tables, for one block size and every tile precision (HIP events,
back-to-back calls) + a digest of the partial rows.  python tools/micro/attn_time.py [block size] [label]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hept_amd import ops
from hept_amd.synthetic import make_inputs

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
label = sys.argv[2] if len(sys.argv) > 2 else "default"
dev = torch.device("cuda:0")
inp = make_inputs([60000], block_size=bs, n_hashes=3, seed=0)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
for prec in ("bf16", "mixed16", "fp32"):
    r = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision=prec)
    qpos, kpos = ops.sort_tables(r["qproj"], r["kproj"], g["combined_shifts"], r["minmax"])
    part = ops.block_attn(r["qhat"], r["kvhat"], qpos, kpos, 24, bs)
    digest = hashlib.sha256(part.cpu().numpy().tobytes()).hexdigest()[:12]
    best = 1e9
    for rep in range(3):
        for _ in range(10):
            ops.block_attn(r["qhat"], r["kvhat"], qpos, kpos, 24, bs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.block_attn(r["qhat"], r["kvhat"], qpos, kpos, 24, bs)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    print(f"{label} B={bs} {prec}: {best:.1f} us per block_attn, part sha {digest}", flush=True)
