"""The oracle's src variant against golden vectors captured from the real reference
(``src/models/attention/hept.py`` with the caller preparation of ``src/models/baselines/transformer.py:43-57``,
imported in the build container by tests/golden/make_golden_src.py), and the host mirror of that preparation.
"""
import numpy as np
import pytest
import torch

import cases
import hept_oracle as ho
from hept_amd.prep import prepare_input_src

SRC = list(cases.SRC_CASES)


def _geo(inp):
    return dict(raw_size=inp["raw_size"], region_indices=(inp["eta_idx"], inp["phi_idx"]), regions_h=inp["regions_h"])


def _oracle(inp, **kw):
    return ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], None, inp["w_rpe_weight"], inp["alpha"],
                      inp["out_weight"], inp["out_bias"], block_size=inp["block_size"], w_per_dist=inp["w_per_dist"],
                      geo=_geo(inp), **kw)


@pytest.mark.parametrize("name", SRC)
def test_src_inputs_rebuild_exactly(name):
    inp, fx = cases.load_case_src(name)
    np.testing.assert_allclose(cases.input_checksums(inp), fx["input_checksums"], rtol=1e-12, atol=1e-9)
    n = inp["q"].shape[0]
    assert n % inp["block_size"] == 0 and 0 <= n - inp["raw_size"] < inp["block_size"]
    assert bool((inp["coords"][inp["raw_size"]:] == 0).all())


@pytest.mark.parametrize("name", SRC)
def test_src_oracle_bit_exact_with_reference_permutations(name):
    inp, fx = cases.load_case_src(name)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int64))
    kp = torch.from_numpy(fx["k_positions"].astype(np.int64))
    res = _oracle(inp, q_positions=qp, k_positions=kp)
    assert torch.equal(res["out"], torch.from_numpy(fx["out"]))
    assert torch.equal(res["hash_span"], torch.from_numpy(fx["hash_span"]))
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    for key in ("q_hashed", "k_hashed", "q_keys", "k_keys"):
        assert torch.equal(res[key][..., rows], torch.from_numpy(fx[key + "_rows"])), key
    assert torch.equal(res["denom"].squeeze(-1)[..., rows], torch.from_numpy(fx["denom_rows"]))
    assert torch.equal(res["per_head"][:, rows], torch.from_numpy(fx["per_head_rows"]))
    if "q_keys" in fx:
        assert torch.equal(res["q_keys"], torch.from_numpy(fx["q_keys"]))
        assert torch.equal(res["k_keys"], torch.from_numpy(fx["k_keys"]))
        assert torch.equal(res["denom"].squeeze(-1), torch.from_numpy(fx["denom"]))


@pytest.mark.parametrize("name", SRC)
def test_src_oracle_own_sort(name):
    """Stable sort of the src keys: padding rows (+inf keys) last in ascending index; the real rows in the
    reference's order up to ties; outputs of the real rows as the reference's."""
    inp, fx = cases.load_case_src(name)
    res = _oracle(inp)
    n, raw = inp["q"].shape[0], inp["raw_size"]
    for pos, ref, keys in ((res["q_positions"], fx["q_positions"], res["q_keys"]),
                           (res["k_positions"], fx["k_positions"], res["k_keys"])):
        assert torch.equal(torch.sort(pos, dim=-1).values, torch.arange(n).expand_as(pos))
        assert torch.equal(pos[..., raw:], torch.arange(raw, n).expand_as(pos[..., raw:]))
        # same sorted key sequence as the reference's (unstable) argsort: they differ inside tie groups only
        ref = torch.from_numpy(ref.astype(np.int64))
        assert torch.equal(torch.gather(keys, -1, pos), torch.gather(keys, -1, ref))
        assert float((pos[..., :raw] == ref[..., :raw]).float().mean()) >= 0.99
    ref_out = torch.from_numpy(fx["out"])
    row_err = (res["out"] - ref_out).abs().amax(-1)
    assert float((row_err <= 1e-5 + 1e-4 * ref_out.abs().amax(-1)).float().mean()) >= 0.99


@pytest.mark.parametrize("name", ["s1_src1000", "s3_src_pileup"])
def test_src_oracle_gradients_equal_reference_autograd(name):
    inp, fx = cases.load_case_src(name)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int64))
    kp = torch.from_numpy(fx["k_positions"].astype(np.int64))
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("q", "k", "v", "w_rpe_weight", "out_weight", "out_bias")}
    res = ho.forward(leaves["q"], leaves["k"], leaves["v"], inp["coords"], None, leaves["w_rpe_weight"], inp["alpha"],
                     leaves["out_weight"], leaves["out_bias"], block_size=inp["block_size"],
                     w_per_dist=inp["w_per_dist"], q_positions=qp, k_positions=kp, keep=False, grad=True,
                     geo=_geo(inp))
    g_out = torch.randn(res["out"].shape, generator=torch.Generator().manual_seed(11))
    res["out"].backward(g_out)
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    assert torch.equal(leaves["q"].grad[rows], torch.from_numpy(fx["ref_dq_rows"]))
    assert torch.equal(leaves["k"].grad[rows], torch.from_numpy(fx["ref_dk_rows"]))
    assert torch.equal(leaves["v"].grad[rows], torch.from_numpy(fx["ref_dv_rows"]))
    assert torch.equal(leaves["w_rpe_weight"].grad, torch.from_numpy(fx["ref_dw_rpe"]))
    assert torch.equal(leaves["out_weight"].grad, torch.from_numpy(fx["ref_dout_w"]))
    # no gradient reaches the padding rows (zero-filled in place by the reference); the fixture recorded the same
    assert float(fx["ref_dq_pad_absmax"]) == 0.0
    raw = inp["raw_size"]
    assert all(float(leaves[x].grad[raw:].abs().max()) == 0.0 for x in ("q", "k", "v"))


def test_prepare_input_src_mirror():
    """Host mirror of src/models/baselines/transformer.py:43-57 rebuilt the fixtures' inputs (the generator asserted
    equality with the reference's own preparation on every real row); here: shapes, padding and region ranges."""
    gen = torch.Generator().manual_seed(3)
    regions = torch.tensor([[[3.0, 4.333333], [5.0, 2.666667]]]).permute(0, 1, 2).contiguous()  # (T=1, 2, H=2)
    x = torch.randn(300, 5, generator=gen)
    coords = torch.randn(300, 4, generator=gen)
    xp, kw = prepare_input_src(x, coords, {"block_size": 128, "regions": regions})
    assert xp.shape == (384, 5) and bool((xp[300:] == 0).all()) and torch.equal(xp[:300], x)
    assert kw["raw_size"] == 300 and kw["coords"].shape == (384, 4) and bool((kw["coords"][300:] == 0).all())
    assert kw["regions_h"].shape == (2, 2)
    eta, phi = kw["region_indices"]
    assert eta.shape == phi.shape == (2, 384) and eta.dtype == torch.float32
    for axis, idx in ((0, eta), (1, phi)):
        for th in range(2):
            width = torch.ceil(kw["regions_h"][axis, th].reciprocal() * 384)
            rank = torch.empty(384, dtype=torch.long)
            key = torch.cat([coords[:, axis], torch.full((84,), float("inf"))])
            rank[torch.sort(key, stable=True).indices] = torch.arange(384)
            assert torch.equal(idx[th], rank // width + 1)
    # a cloud that already is a multiple of the block size is passed through unpadded
    xp2, kw2 = prepare_input_src(x[:256], coords[:256], {"block_size": 128, "regions": regions})
    assert xp2.shape[0] == 256 and kw2["raw_size"] == 256
