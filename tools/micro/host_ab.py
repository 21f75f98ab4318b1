"""Host issue time of one eager forward on a 128-point cloud (host-bound), with and without round 6's two guards (the stream
check of the workspace and the busy flag): same process, same box, alternating.  python tools/micro/host_ab.py"""
import os, sys, time, torch
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


def timed(label):
    run(300); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(2000); t1 = time.perf_counter(); torch.cuda.synchronize()
        best = min(best, (t1 - t0) / 2000 * 1e6)
    print(f"{label}: host issue {best:.1f} us per forward", flush=True)


def old_scratch(self, nbytes, device):
    ws = self._workspace
    if ws is None or ws.numel() < nbytes or ws.device != device:
        ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
        self._workspace = ws
    self._ws_stream_ptr = None
    return ws


new_scratch, new_forward = HEPTAttention._scratch, HEPTAttention.forward
for rep in range(2):
    HEPTAttention._scratch, HEPTAttention.forward = new_scratch, new_forward
    timed("guards on ")
    HEPTAttention._scratch = old_scratch
    HEPTAttention.forward = lambda self, q, k, v, **kwargs: self._forward_impl(q, k, v, kwargs)
    timed("guards off")
