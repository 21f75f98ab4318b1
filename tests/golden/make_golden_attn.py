"""Golden fixtures of the Attn block from the REAL reference ``example/transformer.py:131-165``.

Run once in the build container, after make_golden.py (``python tests/golden/make_golden_attn.py``).  Imports the
reference's ``transformer.Attn`` (PyG stubbed, as in make_golden.py), loads either the shipped checkpoint's layer 0
or a seeded default initialisation, runs it in eval mode on the block input of ``cases.ATTN_CASES`` and stores the
block's weights, its output and the operator's output inside the block (``aggr_out``).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from make_golden import import_reference  # noqa: E402


def main():
    hept, hept_utils, transformer = import_reference()
    ckpt = torch.load("/root/reference/example/ckpt/tracking-60k-model.pt", map_location="cpu", weights_only=False)
    for name, cfg in cases.ATTN_CASES.items():
        inp = cases.build_inputs_attn(name)
        base_cfg = cases.CASES[cfg["base"]]
        torch.manual_seed(cfg.get("seed", 0))
        block = transformer.Attn(base_cfg["coords_dim"], h_dim=cases.H_DIM, num_heads=cases.NUM_HEADS,
                                 block_size=base_cfg["block_size"], n_hashes=base_cfg["n_hashes"],
                                 num_w_per_dist=cases.W_PER_DIST)
        if name == "a1_attn_ckpt6k":
            block.load_state_dict({k[len("attns.0."):]: v for k, v in ckpt.items() if k.startswith("attns.0.")},
                                  strict=True)
            # the trained q/k projections give |q - k|^2 ~ 1e3 on synthetic N(0,1) features: every weight
            # underflows and all denominators sit at the 1e-20 floor.  Scale them as G3 scales q and k, so that
            # the case exercises the attention and not only the layer norms (the dataset is not available here).
            with torch.no_grad():
                block.w_q.weight.mul_(cases.CASES[cfg["base"]]["qk_scale"])
                block.w_k.weight.mul_(cases.CASES[cfg["base"]]["qk_scale"])
        block.eval()
        captured = {}
        block.attn.register_forward_hook(lambda m, a, o: captured.__setitem__("aggr", o.detach().clone()))
        with torch.no_grad():
            y = block(inp["x"], {"coords": inp["coords"], "combined_shifts": inp["combined_shifts"]})
        sd = block.state_dict()
        assert set(sd.keys()) == set(cases.ATTN_KEYS), sorted(sd.keys())
        fx = {"p:" + k: v.numpy() for k, v in sd.items()}
        fx["y"] = y.numpy()
        g = torch.Generator().manual_seed(7)
        rows = torch.randperm(y.shape[0], generator=g)[:256].sort().values
        fx["rows"] = rows.numpy().astype(np.int32)
        fx["aggr_rows"] = captured["aggr"][rows].numpy()
        fx["aggr_abs_mean"] = np.asarray(float(captured["aggr"].abs().mean()))
        t = inp["x"].double().flatten()
        fx["x_checksum"] = np.asarray(float((t * (torch.arange(1, t.numel() + 1, dtype=torch.float64) % 8191)).sum()))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: N={y.shape[0]} |y|mean={y.abs().mean():.4f} |aggr|mean={captured['aggr'].abs().mean():.4f} "
              f"-> {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__":
    main()
