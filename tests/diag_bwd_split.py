"""backward: split-bf16 kernel vs native f32 MFMA kernel (and the oracle's autograd on golden cases); GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import cases
from hept_amd import ops
from hept_amd.synthetic import workload_inputs, WORKLOADS

dev = "cuda:0"


def rel(name, a, b):
    scale = float(b.abs().max()) + 1e-30
    d = (a - b).abs()
    print(f"   {name:10s} max|d|/max|ref| = {float(d.max()) / scale:.3e}   mean|d|/max|ref| = {float(d.mean()) / scale:.3e}", flush=True)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def run(g, B, K):
    H, E, T = g["alpha"].shape
    D = g["q"].shape[1] // H
    C = g["coords"].shape[1]
    sw = ops.rpe_scale(g["w_rpe_weight"], H, D, K)
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], "fp32")
    qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], g["combined_shifts"], ph["minmax"])
    part = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, D, B)
    acc = ops.reduce_tables(part, D).requires_grad_(True)
    out = torch.nn.functional.linear((acc[..., :D] / acc[..., D:D + 1]).reshape(-1, H * D), g["out_weight"], g["out_bias"])
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    out.backward(gout)
    gacc = acc.grad
    a = ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, D, C, B)
    b = ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, D, C, B, f32_mfma=True)
    for nm, x, y in zip(("dq", "dk", "dv", "dcs"), a, b):
        rel(nm, x, y)
    ts = timeit(lambda: ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, D, C, B))
    tm = timeit(lambda: ops.block_attn_bwd(ph["qhat"], ph["kvhat"], qpos, kpos, gacc, D, C, B, f32_mfma=True))
    print(f"   block_attn_bwd + reduce, host-timed: split {ts:.1f} us   mfma {tm:.1f} us", flush=True)


for name in sys.argv[1:] or ["g1_rand512", "g4_pileup", "g6_block100", "g3_ckpt6k", "w:tracking-60k"]:
    if name.startswith("w:"):
        inp = workload_inputs(name[2:], seed=0)
        g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
        print(f"== workload {name[2:]}", flush=True)
        run(g, WORKLOADS[name[2:]]["block_size"], 10)
    else:
        inp, fx = cases.load_case(name)
        g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
        print(f"== {name} N={inp['q'].shape[0]} B={inp['block_size']}", flush=True)
        run(g, inp["block_size"], inp["w_per_dist"])
