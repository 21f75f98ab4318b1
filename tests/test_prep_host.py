"""Host-side prepare_input mirror (hept_amd/prep.py) against the reference's outputs stored in the fixtures.

make_golden.py asserted, in the build container, that the mirror's AND codes equal the reference's
``prepare_input`` codes on every case (differences only at exact coordinate ties, stored as patches);
here the stored reference padding / code checksums are re-checked and the padding rules verified.
"""
import numpy as np
import pytest
import torch

import cases
from hept_amd.prep import bit_shift, get_regions, pad_and_unpad, prepare_input, quantile_partition

PREP_CASES = ["g2_example4k", "g3_ckpt6k", "g4_pileup", "g5_track60k", "g6_block100"]


@pytest.mark.parametrize("name", PREP_CASES)
def test_codes_match_reference_checksum(name):
    inp, fx = cases.load_case(name)
    assert float(inp["combined_shifts"].double().sum()) == float(fx["ref_codes_sum"])
    assert len(fx["code_patch_idx"]) <= 8  # tie-induced only


def _row_digests(codes):
    """(T, H, N) int64 -> (T, H) uint64, as tests/golden/make_golden_codes.py computed them on the reference's codes."""
    import hashlib

    arr = np.ascontiguousarray(codes.numpy().astype("<i8"))
    return np.array([[int.from_bytes(hashlib.sha256(arr[i, j].tobytes()).digest()[:8], "little")
                      for j in range(arr.shape[1])] for i in range(arr.shape[0])], dtype=np.uint64)


@pytest.mark.parametrize("name", PREP_CASES)
def test_codes_equal_the_references_element_for_element(name):
    """Every (table, head) row of the AND codes the tests feed to the operator has the SHA-256 digest of the REFERENCE's
    own ``combined_shifts`` row (stored by make_golden_codes.py from a run of example/transformer.py:35-63): the rows
    are equal element for element, not just in their sum.  The host mirror alone differs only at the stored patches."""
    inp, _ = cases.load_case(name)
    with np.load(cases.os.path.join(cases.GOLDEN_DIR, "ref_codes_digest.npz")) as z:
        want, row_sum, row_max = z[name + "/digest"], z[name + "/row_sum"], z[name + "/row_max"]
    codes = inp["combined_shifts"]
    assert np.array_equal(_row_digests(codes), want)
    assert np.array_equal(codes.sum(-1).numpy(), row_sum) and np.array_equal(codes.amax(-1).numpy(), row_max)
    # the un-patched mirror: rebuilt from the raw inputs, it differs from the reference in <= 8 tie-induced elements
    cfg = cases.CASES[name]
    helper = {"block_size": cfg["block_size"], "num_heads": cases.NUM_HEADS, "regions": inp["regions"]}
    _, kw, _ = prepare_input(torch.arange(inp["n_raw"]), inp["coords_raw"], inp["batch"], helper)
    # (compared on the real points: the mirror draws its own padding duplicates from the same window)
    mine = kw["combined_shifts"][..., inp["unpad_seq"]]
    ref = codes[..., inp["unpad_seq"]]
    assert int((mine != ref).sum()) <= 8


@pytest.mark.parametrize("name", PREP_CASES)
def test_own_padding_follows_reference_rule(name):
    """Pads of cloud i are drawn from the last block_size positions of that cloud in table-0/head-0
    code order; real slots keep their order; the reference's stored pad_seq obeys the same rule."""
    cfg = cases.CASES[name]
    inp, fx = cases.load_case(name)
    B = cfg["block_size"]
    raw = torch.tensor(cfg["cloud_sizes"])
    helper = {"block_size": B, "num_heads": cases.NUM_HEADS, "regions": inp["regions"]}
    n_raw = int(raw.sum())
    pad_seq, kw, unpad = prepare_input(torch.arange(n_raw), inp["coords_raw"], inp["batch"], helper)
    ref_pad = torch.from_numpy(fx["pad_seq"].astype(np.int64))
    assert pad_seq.shape == ref_pad.shape
    assert torch.equal(unpad, inp["unpad_seq"])
    assert torch.equal(pad_seq[unpad], torch.arange(n_raw))
    assert torch.equal(ref_pad[unpad], torch.arange(n_raw))
    # pads: same multiset of (table0, head0) codes as the reference's pads, cloud by cloud
    _, kw_raw, _ = prepare_input(torch.arange(n_raw), inp["coords_raw"], inp["batch"], {**helper, "block_size": 1})
    code00 = kw_raw["combined_shifts"][0, 0]
    mine = torch.sort(code00[pad_seq[~unpad]]).values
    theirs = torch.sort(code00[ref_pad[~unpad]]).values
    assert torch.equal(mine, theirs)
    assert kw["combined_shifts"].shape[-1] % B == 0 and kw["coords"].shape[0] == pad_seq.numel()


def test_quantile_partition_and_bit_shift():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, generator=g)
    regs = torch.tensor([[7.0], [12.3333]])
    out = quantile_partition(torch.argsort(x), regs)
    assert out.shape == (2, 1000) and out.min() == 1
    # monotone in x, equal-population bins of width ceil(n / regions)
    order = torch.argsort(x)
    assert bool((out[:, order][:, 1:] >= out[:, order][:, :-1]).all())
    width = torch.ceil(regs.reciprocal() * 1000)
    assert torch.equal(out[:, order], (torch.arange(1000)[None] // width + 1))
    base = torch.tensor([[1, 5, 3], [2, 2, 9]])
    hi = torch.tensor([[1, 0, 2], [3, 1, 0]])
    packed = bit_shift(base, hi)
    assert torch.equal(packed, torch.tensor([[1 << 3 | 1, 5, 2 << 3 | 3], [3 << 4 | 2, 1 << 4 | 2, 9]]))


def test_get_regions_shape_and_product():
    r = get_regions(150, 3, 8, generator=torch.Generator().manual_seed(0))
    assert r.shape == (3, 2, 8)
    prod = r[:, 0] * r[:, 1]
    assert bool(((prod > 120) & (prod < 185)).all())
    assert torch.allclose(r * 3, torch.round(r * 3))


def test_pad_and_unpad_small_cloud_window():
    # cloud sizes 5 and 3 with block 4: pads = 3 and 1
    sizes = torch.tensor([5, 3])
    batch = torch.repeat_interleave(torch.arange(2), sizes)
    codes = torch.tensor([4, 1, 3, 2, 0, 13, 11, 12])  # cloud id in the high bits
    pad_seq, mask = pad_and_unpad(batch, 4, codes, sizes)
    assert mask.tolist() == [True] * 5 + [False] * 3 + [True] * 3 + [False]
    by_code = torch.sort(codes, stable=True).indices
    assert pad_seq[5:8].tolist() == by_code[1:4].tolist()      # sorted positions [5-4, 5-4+3)
    assert pad_seq[11].item() == by_code[4].item()             # sorted position 8-4
