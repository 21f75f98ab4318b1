"""GPU parity of the reference's src variant (-m gpu): HIP path through the C ABI against the
oracle and the golden vectors captured from the real ``src/models/attention/hept.py``.  Tolerances as in
test_gpu_parity.py (fp32 tiles atol 1e-5 / rtol 1e-4 on >= 99.9 % of the rows, 3x hard bound; 16-bit modes
row-scaled)."""
import numpy as np
import pytest
import torch

import cases
import hept_oracle as ho
from hept_amd import HEPTAttention, ops
from hept_amd.prep import prepare_input_src

pytestmark = pytest.mark.gpu

SRC = list(cases.SRC_CASES)
REL16 = {"bf16": 2.5e-2, "mixed16": 1.0e-2}


def _geo(inp):
    return dict(raw_size=inp["raw_size"], region_indices=(inp["eta_idx"], inp["phi_idx"]), regions_h=inp["regions_h"])


def _oracle(inp, **kw):
    return ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], None, inp["w_rpe_weight"], inp["alpha"],
                      inp["out_weight"], inp["out_bias"], block_size=inp["block_size"], w_per_dist=inp["w_per_dist"],
                      geo=_geo(inp), **kw)


def _gpu(inp, dev):
    return {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}


def _rows_ok(out, ref, atol, rtol=1e-4):
    err = (out - ref).abs()
    return ((err <= atol + rtol * ref.abs()).all(dim=-1)).float().mean().item()


def _rows_scaled_ok(out, ref, rel):
    err = (out - ref).abs().amax(-1)
    return (err <= rel * (ref.abs().amax(-1) + 1e-3)).float().mean().item()


@pytest.mark.parametrize("name", SRC)
def test_src_keys_and_sort_bit_exact(name, gpu_device):
    """Hashes of the real rows, +inf hashes of the padding rows, and the permutation = the stable sort of the
    reference's keys, bit for bit (the keys themselves never leave the kernel; the sort order pins them)."""
    inp, fx = cases.load_case_src(name)
    g = _gpu(inp, gpu_device)
    h, e, t = inp["alpha"].shape
    n, raw, d = inp["q"].shape[0], inp["raw_size"], 24
    want = _oracle(inp)
    sw = ops.rpe_scale(g["w_rpe_weight"], h, d, inp["w_per_dist"])
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], None, "fp32", raw_size=raw)
    qproj, kproj = ph["qproj"].cpu(), ph["kproj"].cpu()
    assert bool(torch.isinf(qproj[..., raw:]).all()) and bool(torch.isinf(kproj[..., raw:]).all())
    torch.testing.assert_close(qproj[..., :raw], want["q_hashed"][..., :raw], rtol=1e-5, atol=1e-5)
    eta, phi, cfac = ops.geo_args((g["eta_idx"], g["phi_idx"]), g["regions_h"], t, h, n)
    qpos, kpos = ops.sort_tables_src(ph["qproj"], ph["kproj"], eta, phi, cfac, ph["minmax"])
    # the kernel's own keys: same formula in torch on the GPU-produced hashes and range
    mm = ph["minmax"].cpu()
    span = (mm[..., 1].amax(-1) - mm[..., 0].amin(-1))[..., None]
    shift = ho.geo_shift(inp["regions_h"], span, (inp["eta_idx"], inp["phi_idx"]), t)
    for proj, pos in ((qproj, qpos), (kproj, kpos)):
        keys = proj + shift
        assert torch.equal(pos.cpu().long(), torch.sort(keys, dim=-1, stable=True).indices)
    # and against the reference's permutation wherever its keys carry no ties
    ref_q = torch.from_numpy(fx["q_positions"].astype(np.int64))
    assert float((qpos.cpu().long()[..., :raw] == ref_q[..., :raw]).float().mean()) >= 0.98


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed16"])
@pytest.mark.parametrize("name", SRC)
def test_src_forward_vs_reference_golden(name, precision, gpu_device):
    inp, fx = cases.load_case_src(name)
    g = _gpu(inp, gpu_device)
    out = ops.forward_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"],
                          inp["raw_size"], g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"],
                          block_size=inp["block_size"], w_per_dist=inp["w_per_dist"], precision=precision).cpu()
    ref = torch.from_numpy(fx["out"])
    raw = inp["raw_size"]
    assert bool(torch.isfinite(out).all())
    if precision == "fp32":
        assert _rows_ok(out[:raw], ref[:raw], 1e-5) >= 0.99      # ties of the reference's unstable argsort
        assert _rows_ok(out[:raw], _oracle(inp)["out"][:raw], 1e-5) >= 0.999
        assert _rows_ok(out[:raw], _oracle(inp)["out"][:raw], 3e-5, 3e-4) >= 0.995
    else:
        assert _rows_scaled_ok(out[:raw], ref[:raw], REL16[precision]) >= 0.99
    # padding rows: zero q^, k^ -> every weight exp(0) = 1 inside the (all-padding tail of the) last block;
    # the reference computes them too (callers slice them away), so they have to agree as well
    if raw < out.shape[0]:
        tol = 1e-4 if precision == "fp32" else 3e-2
        torch.testing.assert_close(out[raw:], ref[raw:], rtol=tol, atol=tol)


def test_src_partial_tables_sum_to_the_whole(gpu_device):
    inp, _ = cases.load_case_src("s1_src1000")
    g = _gpu(inp, gpu_device)
    args = (g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"], inp["raw_size"],
            g["w_rpe_weight"], g["alpha"])
    kw = dict(block_size=inp["block_size"], w_per_dist=inp["w_per_dist"])
    whole = ops.forward_src(*args, g["out_weight"], g["out_bias"], **kw)
    acc = sum(ops.forward_partial_src(*args, t0=t, tl=1, **kw) for t in range(3))
    out = ops.combine_out(acc[None].contiguous(), 24, g["out_weight"], g["out_bias"])
    torch.testing.assert_close(out, whole, rtol=1e-5, atol=1e-6)
    two = ops.forward_partial_src(*args, t0=0, tl=2, **kw) + ops.forward_partial_src(*args, t0=2, tl=1, **kw)
    torch.testing.assert_close(two, acc, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("raw,block,feat", [(1000, 128, 24), (5000, 100, 7), (4096, 128, 3), (60000, 128, 24), (37, 64, 1)])
def test_prepare_input_src_one_hip_call_is_bit_exact(raw, block, feat, gpu_device):
    """hept_prepare_input_src (pad x / coords, two stable ranks with the padding last, float region ids of every
    (table, head), zeroed padding coordinates) against the torch implementation of the same function on CPU tensors,
    which tests/test_oracle_src_golden.py pins on the reference's own kwargs.  Duplicate coordinates included."""
    from hept_amd.prep import get_regions

    gen = torch.Generator().manual_seed(raw)
    x = torch.randn(raw, feat, generator=gen)
    coords = torch.randn(raw, 4, generator=gen)
    coords[::7, 0] = coords[0, 0]          # ties: broken by ascending index on both sides
    coords[1::5, 1] = 0.25
    regions = get_regions(140, 3, 8, generator=gen)
    hp = {"block_size": block, "regions": regions}
    xc, kc = prepare_input_src(x, coords, hp)
    xg, kg = prepare_input_src(x.to(gpu_device), coords.to(gpu_device), {"block_size": block, "regions": regions.to(gpu_device)})
    assert kg["raw_size"] == kc["raw_size"] == raw
    assert torch.equal(xg.cpu(), xc) and torch.equal(kg["coords"].cpu(), kc["coords"])
    assert torch.equal(kg["regions_h"].cpu(), kc["regions_h"])
    for a in (0, 1):
        assert kg["region_indices"][a].dtype == torch.float32
        assert torch.equal(kg["region_indices"][a].cpu(), kc["region_indices"][a].float())


@pytest.mark.parametrize("name", ["s1_src1000", "s3_src_pileup"])
def test_src_module_forward_backward(name, gpu_device):
    """nn.Module with the src variant's kwargs, built by the GPU-side prepare_input_src; inference and training."""
    inp, fx = cases.load_case_src(name)
    dev = gpu_device
    h, e, t = inp["alpha"].shape
    m = HEPTAttention(e, variant="src", h_dim=24, num_heads=h, block_size=inp["block_size"], n_hashes=t,
                      num_w_per_dist=10)
    sd = {"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"],
          "e2lsh.beta": torch.zeros(1, t)}
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    raw = inp["raw_size"]
    # caller-side preparation on the GPU (stable sort through the library): same kwargs as the fixture's
    _, kw = prepare_input_src(torch.zeros(raw, 1, device=dev), inp["coords_raw"].to(dev),
                              {"block_size": inp["block_size"], "regions": inp["regions"].to(dev)})
    assert torch.equal(kw["coords"].cpu(), inp["coords"])
    assert torch.equal(kw["region_indices"][0].cpu(), inp["eta_idx"])
    assert torch.equal(kw["region_indices"][1].cpu(), inp["phi_idx"])
    q, k, v = (inp[x].to(dev).requires_grad_(True) for x in ("q", "k", "v"))
    with torch.no_grad():
        out_inf = m(q.detach(), k.detach(), v.detach(), w_rpe=w_rpe, pe=kw["coords"], **kw)
    ref = torch.from_numpy(fx["out"])
    assert _rows_ok(out_inf.cpu()[:raw], ref[:raw], 1e-5) >= 0.99
    out = m(q, k, v, w_rpe=w_rpe, pe=kw["coords"], **kw)
    torch.testing.assert_close(out.detach(), out_inf, rtol=1e-4, atol=1e-5)
    g_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(11))
    out.backward(g_out.to(dev))
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    for got, key in ((q.grad, "ref_dq_rows"), (k.grad, "ref_dk_rows"), (v.grad, "ref_dv_rows")):
        want = torch.from_numpy(fx[key])
        err = (got.cpu()[rows] - want).abs().amax(-1)
        assert (err <= 2e-4 * float(want.abs().max())).float().mean() >= 0.98
        assert float(got[raw:].abs().max()) == 0.0 if raw < got.shape[0] else True
    want = torch.from_numpy(fx["ref_dw_rpe"])
    assert float((w_rpe.weight.grad.cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
    want = torch.from_numpy(fx["ref_dout_w"])
    assert float((m.out_linear.weight.grad.cpu() - want).abs().max()) <= 5e-3 * float(want.abs().max())


def test_src_training_with_bf16_tiles(gpu_device):
    """The src variant with ``train_tiles = "bf16"``: padding rows (at and after raw_size) are zero-filled by the row
    builder in 16-bit rows too and receive zero gradients; real rows are within 0.15 of the reference's stored gradient
    rows' scale."""
    inp, fx = cases.load_case_src("s1_src1000")
    dev = gpu_device
    h, e, t = inp["alpha"].shape
    m = HEPTAttention(e, variant="src", h_dim=24, num_heads=h, block_size=inp["block_size"], n_hashes=t,
                      num_w_per_dist=10)
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                       "e2lsh.alpha": inp["alpha"], "e2lsh.beta": torch.zeros(1, t)}, strict=True)
    m = m.to(dev).train()
    m.train_tiles = "bf16"
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    raw = inp["raw_size"]
    _, kw = prepare_input_src(torch.zeros(raw, 1, device=dev), inp["coords_raw"].to(dev),
                              {"block_size": inp["block_size"], "regions": inp["regions"].to(dev)})
    q, k, v = (inp[x].to(dev).requires_grad_(True) for x in ("q", "k", "v"))
    out = m(q, k, v, w_rpe=w_rpe, pe=kw["coords"], **kw)
    ref = torch.from_numpy(fx["out"])
    assert float((out.detach().cpu()[:raw] - ref[:raw]).abs().max()) <= 0.1 * float(ref.abs().max())
    out.backward(torch.randn(out.shape, generator=torch.Generator().manual_seed(11)).to(dev))
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    for got, key in ((q.grad, "ref_dq_rows"), (k.grad, "ref_dk_rows"), (v.grad, "ref_dv_rows")):
        want = torch.from_numpy(fx[key])
        assert float((got.cpu()[rows] - want).abs().max()) <= 0.15 * float(want.abs().max()), key
        if raw < got.shape[0]:
            assert float(got[raw:].abs().max()) == 0.0


def test_src_argument_errors(gpu_device):
    inp, _ = cases.load_case_src("s1_src1000")
    g = _gpu(inp, gpu_device)
    kw = dict(block_size=inp["block_size"], w_per_dist=inp["w_per_dist"])
    with pytest.raises(RuntimeError):   # raw_size beyond the padded length
        ops.forward_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"], 5000,
                        g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"], **kw)
    with pytest.raises(ValueError):     # region indices of the wrong shape
        ops.forward_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"][:, :-1], g["phi_idx"]), g["regions_h"],
                        1000, g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"], **kw)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_src_full_size_all_rows_vs_oracle(precision, gpu_device):
    """src variant at tracking-60k size (60 000 real points + 32 padding rows, block 128, 3 tables): every real row
    against the oracle."""
    from hept_amd.synthetic import make_inputs_src

    inp = make_inputs_src(60000, block_size=128, n_hashes=3, seed=4)
    inp["block_size"], inp["w_per_dist"] = 128, 10
    g = _gpu(inp, gpu_device)
    raw = inp["raw_size"]
    out = ops.forward_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"], raw,
                          g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"], block_size=128, w_per_dist=10,
                          precision=precision).cpu()
    kw = dict(tile_dtype=torch.bfloat16) if precision == "bf16" else {}
    want = _oracle(inp, keep=False, **kw)["out"]
    assert bool(torch.isfinite(out).all())
    if precision == "fp32":
        assert _rows_ok(out[:raw], want[:raw], 1e-5) >= 0.995
    else:
        assert _rows_ok(out[:raw], want[:raw], atol=5e-3, rtol=8e-3) >= 0.995


@pytest.mark.parametrize("raw", [3000, 2900])
def test_src_block_256_fp32_value_rows_read_in_place(raw, gpu_device):
    """f32 rows with blocks above 128 points (round 5): the block attention stages its value planes from the caller's
    v instead of from kvhat rows nobody builds any more -- here on the src variant, whose padding rows (>= raw_size)
    must count as v = 0 with the denominator's 1.0, against the oracle on every real row and, bit for bit, against
    the same call composed from the public stages (which build and read the kvhat rows)."""
    from hept_amd.synthetic import make_inputs_src

    inp = make_inputs_src(raw, block_size=256, n_hashes=2, seed=9)
    inp["block_size"], inp["w_per_dist"] = 256, 10
    g = _gpu(inp, gpu_device)
    assert g["q"].shape[0] % 256 == 0 and g["q"].shape[0] > raw
    out = ops.forward_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"], raw,
                          g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"], block_size=256, w_per_dist=10,
                          precision="fp32")
    want = _oracle(inp, keep=False)["out"]
    assert bool(torch.isfinite(out).all())
    assert _rows_ok(out[:raw].cpu(), want[:raw], 1e-5) >= 0.995
    # the composed path: partial rows of each table through the public stages (kvhat rows with their v half), summed
    acc = ops.forward_partial_src(g["q"], g["k"], g["v"], g["coords"], (g["eta_idx"], g["phi_idx"]), g["regions_h"], raw,
                                  g["w_rpe_weight"], g["alpha"], t0=0, tl=2, block_size=256, w_per_dist=10,
                                  precision="fp32")
    n, h = g["q"].shape[0], 8
    sq = ops.rpe_scale(g["w_rpe_weight"], h, 24, 10)
    eta, phi, cfac = ops.geo_args((g["eta_idx"], g["phi_idx"]), g["regions_h"], 2, h, n)
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sq, g["alpha"], None, "fp32", raw_size=raw)
    qp, kp = ops.sort_tables_src(ph["qproj"], ph["kproj"], eta, phi, cfac, ph["minmax"])
    part = ops.block_attn(ph["qhat"], ph["kvhat"], qp, kp, 24, 256)
    assert torch.equal(ops.reduce_tables(part, 24), acc)
