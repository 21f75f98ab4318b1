"""The oracle's Attn block against golden vectors from the real reference
(``example/transformer.py:131-165`` run in eval mode by tests/golden/make_golden_attn.py)."""
import numpy as np
import pytest
import torch

import cases
import hept_oracle as ho


@pytest.mark.parametrize("name", list(cases.ATTN_CASES))
def test_attn_block_oracle_matches_reference(name):
    inp, fx = cases.load_case_attn(name)
    t = inp["x"].double().flatten()
    chk = float((t * (torch.arange(1, t.numel() + 1, dtype=torch.float64) % 8191)).sum())
    np.testing.assert_allclose(chk, float(fx["x_checksum"]), rtol=1e-12)
    res = ho.attn_block(inp["x"], inp["coords"], inp["combined_shifts"], inp["params"], num_heads=8,
                        block_size=inp["block_size"], w_per_dist=inp["w_per_dist"], stable_sort=False, keep=False)
    # same ATen ops in the same order as the reference module, including its (default, unstable) argsort
    assert torch.equal(res["y"], torch.from_numpy(fx["y"]))
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    assert torch.equal(res["aggr"][rows], torch.from_numpy(fx["aggr_rows"]))
    # with the oracle's stable sort only tie groups move
    res2 = ho.attn_block(inp["x"], inp["coords"], inp["combined_shifts"], inp["params"], num_heads=8,
                         block_size=inp["block_size"], w_per_dist=inp["w_per_dist"], keep=False)
    ref = torch.from_numpy(fx["y"])
    bad = (res2["y"] - ref).abs().amax(-1) > 1e-5 + 1e-4 * ref.abs().amax(-1)
    # a swapped tie at a block boundary changes every query of the two blocks it touches: all rows outside such
    # blocks must agree
    touched = torch.zeros(ref.shape[0], dtype=torch.bool)
    b = inp["block_size"]
    for key in ("q_positions", "k_positions"):
        diff = (res[key] != res2[key]).reshape(*res[key].shape[:2], -1, b).any(-1)          # (T, H, blocks)
        for t, h, blk in diff.nonzero().tolist():
            touched[res["q_positions"][t, h, blk * b:(blk + 1) * b]] = True
            touched[res2["q_positions"][t, h, blk * b:(blk + 1) * b]] = True
    assert not bool((bad & ~touched).any())
    assert float(bad.float().mean()) <= 0.10  # (swaps of replicated padding points touch blocks but change nothing)


def test_attn_state_dict_names_match_reference_checkpoint():
    """hept_amd.Attn carries exactly the reference block's state-dict keys (strict load of attns.{i}.*)."""
    from hept_amd import Attn

    inp, _ = cases.load_case_attn("a1_attn_ckpt6k")
    blk = Attn(6, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, n_layers=4, num_regions=150)
    blk.load_state_dict(inp["params"], strict=True)
    assert set(blk.state_dict().keys()) == set(cases.ATTN_KEYS)
