"""Time the table-sharded step on ONE GPU with the collectives forced on (1-rank RCCL group): what the exchange
costs in launches and local copies, before any xGMI traffic.  python tools/shard_overhead.py [bf16|fp32]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.sharding import TableSharding  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tables = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
inp = workload_inputs("tracking-60k", seed=0, n_hashes=tables)
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
C = inp["coords"].shape[1]
w_rpe = torch.nn.Linear(50, 192).to(dev)
with torch.no_grad():
    w_rpe.weight.copy_(g["w_rpe_weight"])
kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
# ".../p2p+view": TableSharding(out_view=True) -- no copy of the gathered output out of the exchange buffer
MODES = (None, "all_to_all/1/p2p", "all_to_all/1/p2p+view", "all_to_all/2/p2p", "all_to_all/4/p2p", "all_to_all/8/p2p", "all_to_all/1/rccl", "all_to_all/2/rccl",
         "all_to_all/2/torch", "reduce_scatter", "all_reduce")
if os.environ.get("HEPT_SHARD_MODES"):   # a subset, comma separated ("plain" = no sharding)
    MODES = tuple(None if m == "plain" else m for m in os.environ["HEPT_SHARD_MODES"].split(","))
for mode in MODES:
    mode, _, rest = (mode or "").partition("/")
    groups, _, via = rest.partition("/")
    via, _, view = via.partition("+")
    mode = mode or None
    m = HEPTAttention(24 + C, h_dim=24, num_heads=8, block_size=128, n_hashes=tables, num_w_per_dist=10, precision=prec,
                      process_group=dist.group.WORLD if mode else None)
    if mode:
        m.sharding = TableSharding(tables, dist.group.WORLD, mode=mode, always_exchange=True,
                                   head_groups=int(groups) if groups else None, out_view=bool(view))
        if via:
            m.sharding.exchange = via
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                       "e2lsh.alpha": inp["alpha"]})
    m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(20):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.synchronize()
    if mode:
        m.sharding.check()
    print(f"{prec} T={tables} mode={mode}{'/' + rest if rest else ''}: "
          f"{(time.perf_counter() - t0) / 200 * 1e6:.1f} us/step", flush=True)
dist.destroy_process_group()
