"""Forward latency at tracking-60k for several block sizes (64, 96, 100 = the reference's yaml, 128), both tile
precisions, with the block_attn share from the HIP-event stage timer."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hept_amd import HEPTAttention, ops
from hept_amd.synthetic import make_inputs
dev = torch.device("cuda", 0)
for b in (100, 128, 96, 64):
    inp = make_inputs([60000], block_size=b, n_hashes=3, seed=0)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    for prec in ("bf16", "fp32"):
        m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=b, n_hashes=3, num_w_per_dist=10, precision=prec).to(dev).eval()
        w_rpe = torch.nn.Linear(50, 192).to(dev)
        kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
        with torch.no_grad():
            for _ in range(10): m(g["q"], g["k"], g["v"], **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): m(g["q"], g["k"], g["v"], **kw)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        ops.profile_enable(2, 50)
        with torch.no_grad():
            for _ in range(50): m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        ms, cnt = ops.profile_read(); ops.profile_enable(0)
        print(f"B={b} {prec}: {dt*1e6:.1f} us/forward  attn {ms['block_attn']/cnt*1e3:.1f} us", flush=True)
