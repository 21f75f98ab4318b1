"""A/B of the two forms of the Attn block's front end (prep_fused): outputs bit-identical?  python tools/micro/fused_ab.py
(run twice: HEPT_FUSED_ROLE_SPLIT=1 selects the first form; this script writes / compares gpurun_out/fused_ab_*.pt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hept_amd import ops
from hept_amd.synthetic import workload_inputs
dev = torch.device("cuda:0")
tag = "old" if os.environ.get("HEPT_FUSED_ROLE_SPLIT") else "new"
out = {}
for wl, prec in (("tracking-60k", "bf16"), ("tracking-6k", "fp32"), ("pileup-8clouds", "mixed16")):
    inp = workload_inputs(wl, seed=0)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    torch.manual_seed(1)
    n = g["q"].shape[0]
    x = torch.randn(n, 24, device=dev)
    lw, lb = torch.randn(24, device=dev), torch.randn(24, device=dev)
    wq, wk, wv = (torch.randn(192, 24, device=dev) * 0.3 for _ in range(3))
    sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
    r = ops.prep_hash_fused(x, lw, lb, 1e-5, wq, wk, wv, g["coords"], sw, g["alpha"], g["combined_shifts"], precision=prec)
    mm = r["minmax"]
    red = torch.stack([mm[..., 0].amin(-1), mm[..., 1].amax(-1), mm[..., 2].amax(-1)])   # the sort only sees the reduction
    for k in ("qhat", "kvhat", "qproj", "kproj"):
        out[f"{wl}/{prec}/{k}"] = r[k].view(torch.uint8 if r[k].dtype != torch.float32 else torch.float32).cpu()
    out[f"{wl}/{prec}/range"] = red.cpu()
os.makedirs("gpurun_out", exist_ok=True)
torch.save(out, f"gpurun_out/fused_ab_{tag}.pt")
other = f"gpurun_out/fused_ab_{'new' if tag == 'old' else 'old'}.pt"
if os.path.exists(other):
    o = torch.load(other)
    bad = [k for k in out if not torch.equal(out[k].view(torch.uint8), o[k].view(torch.uint8))]
    print("bit-identical" if not bad else f"DIFFER: {bad}")
