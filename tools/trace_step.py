"""A few forwards of one configuration, for `rocprofv3 --kernel-trace` timelines (tools/trace_summary.py reads the CSV).
python tools/trace_step.py [plain|p2p|p2pview|rccl|torch] [groups] [steps]   (p2pview: TableSharding(out_view=True))"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import HEPTAttention  # noqa: E402
from hept_amd.sharding import TableSharding  # noqa: E402
from hept_amd.synthetic import workload_inputs  # noqa: E402

how = sys.argv[1] if len(sys.argv) > 1 else "plain"
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
prec = os.environ.get("HEPT_TRACE_PREC", "bf16")
dev = torch.device("cuda", 0)
group = None
if how != "plain":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    group = dist.group.WORLD
wl = os.environ.get("HEPT_TRACE_WORKLOAD", "tracking-60k")
inp = workload_inputs(wl, seed=0, **({"n_hashes": int(os.environ["HEPT_TRACE_TABLES"])} if os.environ.get("HEPT_TRACE_TABLES") else {}))
from hept_amd.synthetic import WORKLOADS  # noqa: E402
g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], 192).to(dev)
with torch.no_grad():
    w_rpe.weight.copy_(g["w_rpe_weight"])
T = inp["alpha"].shape[2]
m = HEPTAttention(inp["alpha"].shape[1], h_dim=24, num_heads=8, block_size=WORKLOADS[wl]["block_size"], n_hashes=T,
                  num_w_per_dist=10, precision=prec, process_group=group)
if group is not None:
    m.sharding = TableSharding(T, group, mode="all_to_all", always_exchange=True, head_groups=groups,
                               out_view=how == "p2pview")
    if how in ("torch", "rccl"):
        m.sharding.exchange = how
m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
m = m.to(dev).eval()
with torch.no_grad():
    for _ in range(steps):
        m(g["q"], g["k"], g["v"], w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    torch.cuda.synchronize()
if group is not None:
    dist.destroy_process_group()
