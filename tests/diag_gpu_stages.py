"""Stage-by-stage GPU-vs-oracle error report (diagnostic; run on the GPU box).  Not a test and not product code."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import hept_oracle as ho
import cases
from hept_amd import ops

dev = "cuda:0"

def stats(name, a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    d = (a - b).abs()
    rel = d / (b.abs() + 1e-30)
    print(f"   {name:28s} max_abs={d.max().item():.3e} mean_abs={d.mean().item():.3e} ref_absmean={b.abs().mean().item():.3e} frac>1e-4={(d>1e-4).float().mean().item():.4f}", flush=True)

def run_case(name, precision):
    inp, fx = cases.load_case(name)
    B, K = inp["block_size"], inp["w_per_dist"]
    print(f"== {name} N={inp['q'].shape[0]} B={B} T={inp['alpha'].shape[2]} precision={precision}", flush=True)
    tile = torch.float32 if precision == "fp32" else torch.bfloat16
    qk = torch.float16 if precision == "mixed16" else None
    orc = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"], inp["alpha"],
                     inp["out_weight"], inp["out_bias"], block_size=B, w_per_dist=K, tile_dtype=tile, qk_dtype=qk)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    H, E, T = inp["alpha"].shape; D = inp["q"].shape[1] // H; N = inp["q"].shape[0]
    sw = ops.rpe_scale(g["w_rpe_weight"], H, D, K)
    stats("sqrt_w", sw, orc["sqrt_w"])
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision)
    stats("qproj", ph["qproj"], orc["q_hashed"]); stats("kproj", ph["kproj"], orc["k_hashed"])
    qh = ph["qhat"].float().cpu(); kv = ph["kvhat"].float().cpu()
    vraw = ph["kvhat"][..., 32:].view(torch.bfloat16).float().cpu() if precision == "mixed16" else kv[..., 32:]
    qref = orc["q_hat"]; kref = orc["k_hat"]
    if precision != "fp32":
        qd = torch.float16 if precision == "mixed16" else torch.bfloat16
        qref = qref.to(qd).float(); kref = kref.to(qd).float()
    stats("qhat[:E]", qh[..., :E], qref); stats("khat[:E]", kv[..., :E], kref)
    vref = inp["v"].reshape(N, H, D).permute(1, 0, 2)
    if precision != "fp32": vref = vref.to(torch.bfloat16).float()
    stats("v", vraw[..., :D], vref)
    print("   v ones col", vraw[..., D].min().item(), vraw[..., D].max().item(), "pad", vraw[..., D + 1:].abs().max().item())
    mm = ph["minmax"].cpu()
    span = (mm[..., 1].amax(-1) - mm[..., 0].amin(-1))
    stats("hash_span", span, orc["hash_span"].squeeze(-1))
    qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], g["combined_shifts"], ph["minmax"])
    # check: permutation + sortedness on GPU's own keys
    for nm, pos, proj in (("q", qpos, ph["qproj"]), ("k", kpos, ph["kproj"])):
        pos_c = pos.long().cpu()
        isperm = bool((torch.sort(pos_c, -1).values == torch.arange(N)).all())
        span_g = span.to(dev)
        keys = proj + (g["combined_shifts"].float() * span_g[..., None])
        sk = torch.gather(keys, -1, pos.long())
        mono = bool((sk[..., 1:] >= sk[..., :-1]).all())
        st = torch.sort(keys, dim=-1, stable=True).indices
        same = (st == pos.long()).float().mean().item()
        print(f"   sort[{nm}] isperm={isperm} monotone={mono} equal_to_torch_stable_on_gpu_keys={same:.6f}", flush=True)
    print("   sort vs oracle positions equal frac:", (qpos.long().cpu() == orc["q_positions"]).float().mean().item(),
          (kpos.long().cpu() == orc["k_positions"]).float().mean().item(), flush=True)
    # block attention with the ORACLE's permutations injected
    part = ops.block_attn(ph["qhat"], ph["kvhat"], orc["q_positions"].to(dev), orc["k_positions"].to(dev), D, B)
    pc = ops.unpack_part(part).cpu()  # (T,N,H,32)
    numer = pc[..., :D].permute(0, 2, 1, 3); denom = pc[..., D].permute(0, 2, 1)
    stats("numer (inj perm)", numer, orc["numer"]); stats("denom (inj perm)", denom, orc["denom"].squeeze(-1))
    print("   part pad cols absmax", pc[..., D + 1:].abs().max().item())
    out = ops.combine_out(part, D, g["out_weight"], g["out_bias"])
    stats("out (inj perm)", out, orc["out"])
    acc = ops.reduce_tables(part)
    out2 = ops.combine_out(acc, D, g["out_weight"], g["out_bias"])
    stats("out via reduce_tables", out2, out)
    # whole op, own sort
    t0 = time.time()
    o = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"],
                    block_size=B, w_per_dist=K, precision=precision)
    torch.cuda.synchronize()
    stats("out (forward, own sort)", o, orc["out"])
    d = (o.cpu() - orc["out"]).abs().amax(-1)
    print(f"   rows >1e-4: {(d>1e-4).sum().item()} / {N};  rows>1e-2: {(d>1e-2).sum().item()}")
    if "out" in fx:
        stats("out vs REFERENCE golden", o, torch.from_numpy(fx["out"]))

if __name__ == "__main__":
    names = sys.argv[1:] or ["g1_rand512", "g6_block100", "g4_pileup", "g3_ckpt6k"]
    print(torch.cuda.get_device_name(0))
    for nm in names:
        for prec in ("fp32", "bf16", "mixed16"):
            try:
                run_case(nm, prec)
            except Exception as e:
                import traceback; traceback.print_exc()
