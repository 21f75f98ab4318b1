"""f32 tiles: split-bf16 kernel vs native f32 MFMA kernel vs the CPU oracle, same permutations (diagnostic, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import hept_oracle as ho
import cases
from hept_amd import ops
from hept_amd.synthetic import workload_inputs

dev = "cuda:0"


def cmp(name, a, b):
    d = (a.double() - b.double()).abs()
    tol = 1e-5 + 1e-4 * b.double().abs()
    print(f"   {name:34s} max_abs={d.max().item():.3e} mean_abs={d.mean().item():.3e} "
          f"max_rel_to_tol={(d / tol).max().item():.3f} frac_over_tol={(d > tol).double().mean().item():.2e}", flush=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def staged(g, B, K, orc=None):
    H, E, T = g["alpha"].shape
    D = g["q"].shape[1] // H
    sw = ops.rpe_scale(g["w_rpe_weight"], H, D, K)
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], "fp32")
    if orc is None:
        qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], g["combined_shifts"], ph["minmax"])
    else:
        qpos, kpos = orc["q_positions"].to(dev), orc["k_positions"].to(dev)
    ps = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, D, B)
    pm = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, D, B, f32_mfma=True)
    cmp("part numer split vs mfma", ps[..., :D], pm[..., :D])
    cmp("part denom split vs mfma", ps[..., D], pm[..., D])
    os_ = ops.combine_out(ps, D, g["out_weight"], g["out_bias"])
    om = ops.combine_out(pm, D, g["out_weight"], g["out_bias"])
    cmp("out split vs mfma", os_, om)
    if orc is not None:
        numer = orc["numer"].permute(0, 2, 1, 3).to(dev)
        denom = orc["denom"].squeeze(-1).permute(0, 2, 1).to(dev)
        cmp("numer split vs oracle", ps[..., :D], numer)
        cmp("numer mfma  vs oracle", pm[..., :D], numer)
        cmp("denom split vs oracle", ps[..., D], denom)
        cmp("denom mfma  vs oracle", pm[..., D], denom)
        cmp("out split vs oracle", os_.cpu(), orc["out"])
        cmp("out mfma  vs oracle", om.cpu(), orc["out"])
    ts = timeit(lambda: ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, D, B))
    tm = timeit(lambda: ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, D, B, f32_mfma=True))
    print(f"   block_attn host-timed: split {ts:.1f} us   mfma {tm:.1f} us", flush=True)


for name in sys.argv[1:] or ["g3_ckpt6k", "g4_pileup"]:
    if name.startswith("w:"):
        inp = workload_inputs(name[2:], seed=0)
        print(f"== workload {name[2:]}", flush=True)
        g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
        from hept_amd.synthetic import WORKLOADS
        staged(g, WORKLOADS[name[2:]]["block_size"], 10)
        continue
    inp, fx = cases.load_case(name)
    print(f"== {name} N={inp['q'].shape[0]}", flush=True)
    B, K = inp["block_size"], inp["w_per_dist"]
    orc = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"], inp["alpha"],
                     inp["out_weight"], inp["out_bias"], block_size=B, w_per_dist=K)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    staged(g, B, K, orc)
