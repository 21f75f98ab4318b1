"""Fingerprints of the Attn block front end's outputs (prep_hash_fused) on three workloads / precisions: run before and
after a change of the kernel, the lines must not change (the rows, hashes and reduced ranges are defined bit for bit).
python tools/micro/fused_ab.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hept_amd import ops
from hept_amd.synthetic import workload_inputs
dev = torch.device("cuda:0")
for wl, prec in (("tracking-60k", "bf16"), ("tracking-6k", "fp32"), ("pileup-8clouds", "mixed16"), ("tracking-60k-t8", "bf16")):
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
    r["range"] = torch.stack([mm[..., 0].amin(-1), mm[..., 1].amax(-1), mm[..., 2].amax(-1)])   # what the sort sees
    fp = []
    for k in ("qhat", "kvhat", "qproj", "kproj", "range"):
        b = r[k].contiguous().view(torch.uint8).cpu().numpy().tobytes()
        fp.append(f"{k}={hashlib.sha256(b).hexdigest()[:12]}")
    print(f"{wl:16s} {prec:8s} " + " ".join(fp))
