"""Host time to ISSUE one forward (Python + ctypes + 7 kernel launches) vs end-to-end time, for small clouds."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import make_inputs
dev = torch.device("cuda", 0)
for n_raw in (128, 1024, 6000):
    inp = make_inputs([n_raw], block_size=128, n_hashes=3, seed=1)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16").to(dev).eval()
    w_rpe = torch.nn.Linear(50, 192).to(dev)
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        for _ in range(50):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            m(g["q"], g["k"], g["v"], **kw)
        t_issue = (time.perf_counter() - t0) / 1000
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 1000
    print(f"N_raw={n_raw}: host issue {t_issue*1e6:.1f} us/call, end-to-end {t_all*1e6:.1f} us/call")
