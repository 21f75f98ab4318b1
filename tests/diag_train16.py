"""Measures how far the opt-in bf16 training tiles (HEPTAttention.train_tiles = "bf16") are from the fp32 ones on the
golden cases: max error / tensor scale and mean error / mean magnitude, per tensor.  Basis of the bounds in
tests/test_gpu_backward.py::test_bf16_training_tiles.  python tests/diag_train16.py (on a GPU box)."""
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(_HERE), _HERE, os.path.join(_HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import cases  # noqa: E402
from hept_amd import HEPTAttention  # noqa: E402


def run(inp, tiles, dev):
    h, e, t = inp["alpha"].shape
    d = inp["q"].shape[1] // h
    m = HEPTAttention(e, h_dim=d, num_heads=h, block_size=inp["block_size"], n_hashes=t, num_w_per_dist=10)
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                       "e2lsh.alpha": inp["alpha"]}, strict=True)
    m = m.to(dev).train()
    m.train_tiles = tiles
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    q, k, v = (inp[x].to(dev).requires_grad_(True) for x in ("q", "k", "v"))
    out = m(q, k, v, w_rpe=w_rpe, coords=inp["coords"].to(dev), combined_shifts=inp["combined_shifts"].to(dev))
    out.backward(torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(dev))
    return [x.detach().cpu() for x in (out, q.grad, k.grad, v.grad, w_rpe.weight.grad, m.out_linear.weight.grad)]


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    for name in ("g1_rand512", "g6_block100", "g4_pileup", "g3_ckpt6k", "g5_track60k"):
        inp, _ = cases.load_case(name)
        ref, got = run(inp, "fp32", dev), run(inp, "bf16", dev)
        msg = []
        for nm, a, b in zip(("out", "dq", "dk", "dv", "dw_rpe", "dW_out"), got, ref):
            msg.append("%s %.1e/%.1e" % (nm, float((a - b).abs().max() / b.abs().max()),
                                         float((a - b).abs().mean() / b.abs().mean())))
        print(name, "  ".join(msg), flush=True)
