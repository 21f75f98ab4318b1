"""Row-scaled error statistics of the 16-bit modes against the reference goldens (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, cases
from hept_amd import ops
dev = "cuda:0"
for name in ["g1_rand512", "g2_example4k", "g3_ckpt6k", "g4_pileup", "g6_block100"]:
    inp, fx = cases.load_case(name)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    H, E, T = inp["alpha"].shape; D = 24
    ref = torch.from_numpy(fx["out"])
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(dev); kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(dev)
    for prec in ("bf16", "mixed16"):
        sw = ops.rpe_scale(g["w_rpe_weight"], H, D, 10)
        ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], prec)
        part = ops.block_attn(ph["qhat"], ph["kvhat"], qp, kp, D, inp["block_size"])
        out = ops.combine_out(part, D, g["out_weight"], g["out_bias"]).cpu()
        err = (out - ref).abs()
        scale = ref.abs().amax(-1, keepdim=True)
        rel = (err / (scale + 1e-3)).amax(-1)
        qs = torch.quantile(rel, torch.tensor([0.5, 0.9, 0.99, 0.999]))
        print(f"{name:14s} {prec:8s} row-scaled err quantiles 50/90/99/99.9%: {qs.tolist()}  max {rel.max():.3e}  rowscale mean {scale.mean():.3f}")
