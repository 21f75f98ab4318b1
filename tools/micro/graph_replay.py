"""Small clouds: one forward per HIP-graph replay (torch.cuda.CUDAGraph over the module call) against eager calls --
how much of example-4k / tracking-6k is the host's issue time.  python tools/micro/graph_replay.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.synthetic import WORKLOADS, workload_inputs  # noqa: E402

dev = torch.device("cuda", 0)
for wl in ("example-4k", "tracking-6k", "tracking-60k"):
    for prec in ("fp32", "bf16"):
        inp = workload_inputs(wl, seed=0)
        g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
        T = inp["alpha"].shape[2]
        m = HEPTAttention(inp["alpha"].shape[1], h_dim=24, num_heads=8, block_size=WORKLOADS[wl]["block_size"], n_hashes=T,
                          num_w_per_dist=10, precision=prec)
        m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
        m = m.to(dev).eval()
        w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], 192).to(dev)
        with torch.no_grad():
            w_rpe.weight.copy_(g["w_rpe_weight"])
            kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
            for _ in range(20):
                m(g["q"], g["k"], g["v"], **kw)
            torch.cuda.synchronize()

            def timed(fn, n=300):
                best = 1e9
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / n * 1e6)
                return best

            eager = timed(lambda: m(g["q"], g["k"], g["v"], **kw))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m(g["q"], g["k"], g["v"], **kw)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = m(g["q"], g["k"], g["v"], **kw)
            rep = timed(graph.replay)
        print(f"{wl:14s} {prec}: eager {eager:6.1f} us/forward   graph replay {rep:6.1f} us/forward", flush=True)
