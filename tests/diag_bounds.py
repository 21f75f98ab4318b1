"""Diagnostic (not a test): measures the bounds the parity tests state -- run on the GPU box, prints one line per case.
fp32: the multiplier m such that EVERY element meets |out - ref| <= m * (atol + 1e-4 |ref|); 16-bit: worst row-scaled error."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import hept_oracle as ho  # noqa: E402
from test_gpu_parity import ALL, ATOL, _gpu, _oracle, _staged  # noqa: E402

dev = torch.device("cuda:0")


def mult(out, ref, atol):
    return float(((out - ref).abs() / (atol + 1e-4 * ref.abs())).max())


for name in ALL:
    inp, fx = cases.load_case(name)
    g = _gpu(inp, dev)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(dev)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(dev)
    ref = torch.from_numpy(fx["out"])
    out = _staged(g, inp, "fp32", qp, kp)["out"].cpu()
    print(f"golden {name} fp32: every-element multiplier {mult(out, ref, ATOL.get(name, 1e-5)):.3f}", flush=True)
    for prec in ("bf16", "mixed16"):
        o = _staged(g, inp, prec, qp, kp)["out"].cpu()
        worst = ((o - ref).abs().amax(-1) / (ref.abs().amax(-1) + 1e-3)).max().item()
        print(f"golden {name} {prec}: worst row-scaled error {worst:.3e}", flush=True)

from hept_amd.synthetic import WORKLOADS, make_inputs, workload_inputs  # noqa: E402

full = []
for wl in ("tracking-60k", "pileup-8clouds"):
    inp = workload_inputs(wl, seed=0)
    inp["block_size"], inp["w_per_dist"] = WORKLOADS[wl]["block_size"], 10
    full.append((wl, inp))
inp = make_inputs([60000], block_size=100, n_hashes=3, seed=0)
inp["block_size"], inp["w_per_dist"] = 100, 10
full.append(("60k-block100", inp))
for wl, inp in full:
    g = _gpu(inp, dev)
    st = _staged(g, inp, "fp32")
    qp, kp = st["qpos"].long().cpu(), st["kpos"].long().cpu()
    orc = _oracle(inp, q_positions=qp, k_positions=kp, keep=False)
    out = st["out"].cpu()
    print(f"full {wl} fp32 (GPU perms injected): every-element multiplier {mult(out, orc['out'], 1e-5):.3f}", flush=True)
    own = _oracle(inp, keep=True)
    for nm, pos, theirs in (("q", qp, own["q_positions"]), ("k", kp, own["k_positions"])):
        print(f"full {wl} {nm}: fraction of sorted positions that differ {(pos != theirs).float().mean().item():.3e}", flush=True)
    # attribution: rows that differ end to end vs rows whose block membership differs between the permutations
    b = inp["block_size"]
    bad = ~(((out - own["out"]).abs() <= 1e-5 + 1e-4 * own["out"].abs()).all(-1))
    n = out.shape[0]
    moved = torch.zeros(n, dtype=torch.bool)
    for pos, theirs in ((qp, own["q_positions"]), (kp, own["k_positions"])):
        inv_a = torch.empty_like(pos); inv_a.scatter_(-1, pos, torch.arange(n).expand_as(pos))
        inv_b = torch.empty_like(theirs); inv_b.scatter_(-1, theirs, torch.arange(n).expand_as(theirs))
        blk_diff = (inv_a // b) != (inv_b // b)         # (T, H, N): the point sits in another block
        if pos is qp:
            moved |= blk_diff.any(0).any(0)
            qblk_a = inv_a // b
        else:
            # a query row is affected when ANY key of its block changed: block-level flag, mapped to the queries
            t_, h_, _ = pos.shape
            cnt = torch.zeros(t_, h_, n // b, dtype=torch.int32)
            cnt.scatter_add_(-1, inv_a // b, blk_diff.int())
            cnt.scatter_add_(-1, inv_b // b, blk_diff.int())
            moved |= (cnt > 0).gather(-1, qblk_a).any(0).any(0)
    print(f"full {wl}: {int(bad.sum())} rows differ end to end, {int(moved.sum())} rows touched by a moved point, "
          f"differing rows NOT touched: {int((bad & ~moved).sum())}", flush=True)
