"""Context number for BASELINE.md §1 (whole-model inference latency, 29.96 ms published on an unnamed CUDA GPU):
a model of the reference's shape -- feature encoder, 4 Attn blocks (block_size 100, 3 tables, 8 heads, h_dim 24), the
concatenating linear W and a 5-layer MLP head -- on one synthetic tracking-60k cloud.  The Attn blocks are this
repository's fused blocks; encoder / W / head are plain torch modules of the reference's sizes (the reference's head is
torch_geometric's MLP, not installed here, so this is a latency estimate, not a parity claim).
python tools/model_latency.py [bf16|fp32]"""
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hept_amd import Attn, get_regions, prepare_input  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n_raw, in_dim, coords_dim, h_dim, n_layers = 60000, 12, 6, 24, 4
cfg = dict(h_dim=h_dim, num_heads=8, block_size=100, n_hashes=3, num_w_per_dist=10, n_layers=n_layers, num_regions=150)


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.feat_encoder = nn.Sequential(nn.Linear(in_dim, h_dim), nn.ReLU(), nn.Linear(h_dim, h_dim))
        self.attns = nn.ModuleList([Attn(coords_dim, precision=prec, **cfg) for _ in range(n_layers)])
        self.W = nn.Linear(h_dim * (n_layers + 1), h_dim // 2, bias=False)
        layers, d = [], h_dim // 2
        for i in range(5):  # MLP(in 12, hidden 256, out 12, 5 layers, layer_norm, tanh)
            o = 256 if i < 4 else h_dim // 2
            layers += [nn.Linear(d, o)] + ([nn.LayerNorm(o), nn.Tanh()] if i < 4 else [])
            d = o
        self.mlp_out = nn.Sequential(*layers)
        self.regions = nn.Parameter(get_regions(150, 3, 8), requires_grad=False)

    def forward(self, x, coords, batch):
        x, kwargs, unpad = prepare_input(x, coords, batch, {"block_size": 100, "num_heads": 8, "regions": self.regions})
        enc = self.feat_encoder(x)
        feats = [enc]
        for blk in self.attns:
            enc = blk(enc, kwargs)
            feats.append(enc)
        enc = self.W(torch.cat(feats, dim=-1))
        return (enc + self.mlp_out(enc))[unpad]


m = Model().to(dev).eval()
with torch.no_grad():
    for blk in m.attns:  # synthetic features: scale q/k so that the attention is not degenerate
        blk.w_q.weight.mul_(0.3)
        blk.w_k.weight.mul_(0.3)
x = torch.randn(n_raw, in_dim, device=dev)
coords = torch.randn(n_raw, coords_dim, device=dev)
batch = torch.zeros(n_raw, dtype=torch.long, device=dev)
with torch.no_grad():
    for _ in range(5):
        out = m(x, coords, batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        out = m(x, coords, batch)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
ts.sort()
print(f"{prec}: whole-model-shaped forward on one {n_raw}-point cloud: median {ts[len(ts)//2]*1e3:.3f} ms "
      f"(min {ts[0]*1e3:.3f}); output {tuple(out.shape)}, finite={bool(torch.isfinite(out).all())}")
