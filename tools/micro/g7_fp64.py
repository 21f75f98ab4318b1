"""G7 (shipped layer-0 scales, raw coordinates) against the float64 evaluation of the reference: row errors of the HIP
f32 modes with the reference's permutations injected.  python tools/micro/g7_fp64.py [label]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import cases
from hept_amd import ops
label = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda:0")
for case, atol in (("g7_ckpt_rawcoords", 1e-3), ("g3_ckpt6k", 1e-3)):
    inp, fx = cases.load_case(case)
    if "out_fp64" not in fx:
        continue
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(dev)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(dev)
    o64, o32 = torch.from_numpy(fx["out_fp64"]), torch.from_numpy(fx["out"]).double()
    tol = atol + 1e-4 * o64.abs()
    def rep(name, o):
        e = (o - o64).abs(); r = e.amax(1)
        print(f"{label} {case} {name}: rows in tol {float((e <= tol).all(1).float().mean()):.4f} median {float(r.median()):.3e} "
              f"mean {float(r.mean()):.3e} max {float(r.max()):.3e} rows > 0.5: {int((r > 0.5).sum())}", flush=True)
    rep("reference fp32", o32)
    sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
    for prec, mfma in (("fp32", False), ("fp32", True)):
        ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision=prec)
        part = ops.block_attn(ph["qhat"], ph["kvhat"], qp, kp, 24, inp["block_size"], f32_mfma=mfma)
        out = ops.combine_out(part, 24, g["out_weight"], g["out_bias"]).cpu().double()
        rep("HIP f32 MFMA kernel" if mfma else "HIP split-bf16 kernel", out)
