"""Forward latency / throughput of every BASELINE.json configuration on one GPU (inputs resident in HBM).
python tools/config_sweep.py  -> one line per (workload, precision)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.synthetic import WORKLOADS, workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
ONLY = [w for w in os.environ.get("HEPT_SWEEP_ONLY", "").split(",") if w]   # a subset of the workloads
for name in (ONLY or WORKLOADS):
    inp = workload_inputs(name, seed=0)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    h, e, t = inp["alpha"].shape
    c = inp["coords"].shape[1]
    n, n_raw = inp["q"].shape[0], inp["n_raw"]
    b = WORKLOADS[name]["block_size"]
    w_rpe = torch.nn.Linear((c - 1) * 10, 192).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(g["w_rpe_weight"])
    for prec in ("fp32", "bf16", "mixed16"):
        m = HEPTAttention(e, h_dim=24, num_heads=h, block_size=b, n_hashes=t, num_w_per_dist=10, precision=prec)
        m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                           "e2lsh.alpha": inp["alpha"]})
        m = m.to(dev).eval()
        kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
        with torch.no_grad():
            for _ in range(20):
                m(g["q"], g["k"], g["v"], **kw)
            torch.cuda.synchronize()
            reps, us = 300, float("inf")
            for _ in range(3):  # best of three: the small clouds are bound by the host's launch rate, which is noisy
                t0 = time.perf_counter()
                for _ in range(reps):
                    m(g["q"], g["k"], g["v"], **kw)
                torch.cuda.synchronize()
                us = min(us, (time.perf_counter() - t0) / reps * 1e6)
        flops = 2.0 * t * h * n * b * (e + 24)
        print(f"{name:16s} N_raw={n_raw:6d} N={n:6d} B={b:3d} T={t} C={c} {prec:8s} {us:8.1f} us/forward "
              f"{n_raw / us:8.2f} M points/s  {flops / us * 1e-6:7.1f} TFLOP/s", flush=True)
