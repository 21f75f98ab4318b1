"""Where the host time of one eager forward goes (cProfile over 3000 calls on a 128-point cloud: host-bound)."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hept_amd import HEPTAttention
from hept_amd.synthetic import make_inputs
dev = torch.device("cuda", 0)
inp = make_inputs([128], block_size=128, n_hashes=3, seed=1)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16").to(dev).eval()
w_rpe = torch.nn.Linear(50, 192).to(dev)
kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
def run(n):
    with torch.no_grad():
        for _ in range(n):
            m(g["q"], g["k"], g["v"], **kw)
run(200); torch.cuda.synchronize()
t0 = time.perf_counter(); run(3000); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host issue {(t1 - t0) / 3000 * 1e6:.1f} us per forward")
pr = cProfile.Profile(); pr.enable(); run(3000); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
