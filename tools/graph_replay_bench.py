"""Host-bound regime (small clouds): eager forward vs replay of the same forward captured in a HIP graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda", 0)
for n_raw in (1024, 6000, 60000):
    inp = make_inputs([n_raw], block_size=128, n_hashes=3, seed=1)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10,
                      precision="bf16").to(dev).eval()
    w_rpe = torch.nn.Linear(50, 192).to(dev)
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])

    def run():
        return m(g["q"], g["k"], g["v"], **kw)

    with torch.no_grad():
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            run()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 500
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 500
    print(f"N_raw={n_raw:6d}: eager {eager*1e6:7.1f} us/forward   graph replay {replay*1e6:7.1f} us/forward", flush=True)
