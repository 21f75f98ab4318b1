"""Sanity run far above the benchmark size (480k and 1.92M points in one cloud): finishes, finite, memory use.
The checksums of runs under the measurement switches (HEPT_NO_ROW_RIDERS=1: the row builder writes the v rows itself;
HEPT_NO_STAGED_COMBINE=1: the combine loads its rows lane by lane) must be equal: at these sizes a combine wave takes
several tiles, the only place where the staged rows' cross-tile prefetch runs.
python tools/large_cloud_check.py [precision]"""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hept_amd import HEPTAttention
from hept_amd.synthetic import make_inputs
dev = torch.device("cuda", 0)
torch.manual_seed(0)   # the module parameters: the same in every run
for n_raw in (480000, 1920000):
    inp = make_inputs([n_raw], block_size=128, n_hashes=3, seed=1)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision=(sys.argv[1] if len(sys.argv) > 1 else "bf16")).to(dev).eval()
    with torch.no_grad():
        m.e2lsh.alpha.copy_(g["alpha"])
    w_rpe = torch.nn.Linear(50, 192).to(dev)
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        for _ in range(3):
            out = m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"N_raw={n_raw}: {dt*1e3:.3f} ms/forward, {n_raw/dt/1e6:.1f} M points/s, finite={bool(torch.isfinite(out).all())}, "
          f"mem={torch.cuda.max_memory_allocated()/2**30:.2f} GiB, "
          f"checksum={float(out.double().sum()):.10e} / {float(out.double().abs().sum()):.10e}", flush=True)
    del g, inp, m
    torch.cuda.empty_cache()
