"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle and the
reference's golden vectors.  Tolerances: fp32 tiles with identical permutations
atol 1e-5 / rtol 1e-4 (trained-weight case G3: atol 1e-3, its logits cancel at |q^|^2 ~ 1e3);
bf16 tiles atol 2e-2 against the fp32 reference, tight against the oracle's bf16 model.
"""
import numpy as np
import pytest
import torch

import cases
import hept_oracle as ho
from hept_amd import ops

pytestmark = pytest.mark.gpu

SMALL = ["g1_rand512", "g2_example4k", "g4_pileup", "g6_block100"]
ALL = SMALL + ["g3_ckpt6k"]
ATOL = {"g3_ckpt6k": 1e-3}
# agreement of the 16-bit kernels with the oracle's MODEL of them (same rounded tiles, fp32 accumulate): a
# last-bit fp32 difference can flip the rounding of a stored bf16 numerator, so G3 (|out| ~ 2..8) sits lower
BF16_MODEL_ROWS = {"g3_ckpt6k": 0.96}


def _oracle(inp, **kw):
    return ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                      inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=inp["block_size"],
                      w_per_dist=inp["w_per_dist"], **kw)


def _gpu(inp, dev):
    return {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}


def _dims(inp):
    h, e, t = inp["alpha"].shape
    return inp["q"].shape[0], h, inp["q"].shape[1] // h, e, t


def _forward(g, inp, precision):
    return ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                       g["out_weight"], g["out_bias"], block_size=inp["block_size"], w_per_dist=inp["w_per_dist"],
                       precision=precision)


def _staged(g, inp, precision, qpos=None, kpos=None):
    n, h, d, e, t = _dims(inp)
    sw = ops.rpe_scale(g["w_rpe_weight"], h, d, inp["w_per_dist"])
    ph = ops.prep_hash(g["q"], g["k"], g["v"], g["coords"], sw, g["alpha"], g["combined_shifts"], precision)
    if qpos is None:
        qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], g["combined_shifts"], ph["minmax"])
    part = ops.block_attn(ph["qhat"], ph["kvhat"], qpos, kpos, d, inp["block_size"],
                          f32_mfma={"fp32_mfma": True, "fp32_diff": "diff"}.get(precision, False))
    out = ops.combine_out(part, d, g["out_weight"], g["out_bias"])
    return dict(sqrt_w=sw, part=part, out=out, qpos=qpos, kpos=kpos, **ph)


def _model_kw(precision):
    """Oracle switches that model a HIP precision mode."""
    if precision == "bf16":
        return dict(tile_dtype=torch.bfloat16)
    if precision == "mixed16":
        return dict(tile_dtype=torch.bfloat16, qk_dtype=torch.float16)
    return {}


def _rows_scaled_ok(out, ref, rel):
    """Fraction of rows whose worst element error is <= rel * (largest |reference| of that row).  The 16-bit
    modes round weights, values and stored numerators to bf16: their error scales with the row, not the element."""
    err = (out - ref).abs().amax(-1)
    return (err <= rel * (ref.abs().amax(-1) + 1e-3)).float().mean().item()


# stated tolerances of the 16-bit modes against the fp32 REFERENCE (measured 99th percentiles: bf16 <= 1.9e-2,
# mixed16 <= 0.85e-2 of the row scale on every golden case, tests/diag_tol_probe.py)
REL16 = {"bf16": 2.5e-2, "mixed16": 1.0e-2}
# ... and the bound EVERY row meets, per golden case, set from measurement (tests/diag_bounds.py, round 3; the kernels
# are deterministic, so the measured worst row is reproduced on every box) at <= 2x the measured worst row.  bf16
# rounds q^/k^ to 8 bits, so its worst rows are those of the trained-checkpoint case G3 (|q^|^2 ~ 1e3); fp16 q^/k^
# rows (mixed16) remove that term.          measured worst row:   bf16      mixed16
REL16_ALL_ROWS = {
    "g1_rand512":   {"bf16": 4.0e-2, "mixed16": 1.2e-2},     # 2.30e-2   6.2e-3
    "g2_example4k": {"bf16": 8.0e-2, "mixed16": 1.5e-2},     # 5.04e-2   7.7e-3
    "g3_ckpt6k":    {"bf16": 1.0e-1, "mixed16": 2.5e-2},     # 7.23e-2   1.68e-2
    "g4_pileup":    {"bf16": 1.0e-2, "mixed16": 1.0e-2},     # 5.7e-3    5.2e-3
    "g6_block100":  {"bf16": 1.0e-2, "mixed16": 1.0e-2},     # 5.4e-3    5.4e-3
}
# synthetic full-size clouds (default-initialised weights, |q^| ~ 5): every row within this
REL16_ALL_ROWS_FULL = {"bf16": 1.0e-1, "mixed16": 2.5e-2}
# fp32 tiles, identical permutations: the stated tolerance (atol 1e-5 / rtol 1e-4) holds for EVERY element of every
# golden case (measured worst element: 0.44x of it); at full size (60k points, ~1.4 M elements, a few rows whose total
# weight sits near the 1e-20 denominator floor) >= 99.9 % of rows meet it and EVERY element is within HARD_X times it
# (measured worst: 1.27x, tests/diag_bounds.py)
HARD_X = 2.0


def _rows_ok(out, ref, atol, rtol=1e-4):
    err = (out - ref).abs()
    return ((err <= atol + rtol * ref.abs()).all(dim=-1)).float().mean().item()


@pytest.mark.parametrize("name", ALL)
def test_prep_hash_stage(name, gpu_device):
    inp, _ = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    n, h, d, e, t = _dims(inp)
    orc = _oracle(inp)
    st = _staged(g, inp, "fp32")
    torch.testing.assert_close(st["sqrt_w"].cpu(), orc["sqrt_w"], rtol=2e-6, atol=0)
    scale = float(orc["q_hashed"].abs().max())
    torch.testing.assert_close(st["qproj"].cpu(), orc["q_hashed"], rtol=0, atol=4e-6 * scale)
    torch.testing.assert_close(st["kproj"].cpu(), orc["k_hashed"], rtol=0, atol=4e-6 * scale)
    qh, kv = st["qhat"].cpu(), st["kvhat"].cpu()
    torch.testing.assert_close(qh[..., :e], orc["q_hat"], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(kv[..., :e], orc["k_hat"], rtol=2e-6, atol=1e-7)
    assert torch.equal(kv[..., 32:32 + d], inp["v"].reshape(n, h, d).permute(1, 0, 2))
    assert bool((kv[..., 32 + d] == 1).all()) and bool((kv[..., 33 + d:] == 0).all()) and bool((qh[..., e:31] == 0).all())
    torch.testing.assert_close(qh[..., 31], -0.5 * (orc["q_hat"] ** 2).sum(-1), rtol=1e-5, atol=1e-6)
    mm = st["minmax"].cpu()
    span = mm[..., 1].amax(-1) - mm[..., 0].amin(-1)
    torch.testing.assert_close(span, orc["hash_span"].squeeze(-1), rtol=1e-5, atol=0)


@pytest.mark.parametrize("name", ALL)
def test_sort_is_stable_sort_of_the_keys(name, gpu_device):
    """Bit-exact index work: permutation, keys non-decreasing, ties in ascending index = torch stable sort."""
    inp, _ = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    n = inp["q"].shape[0]
    st = _staged(g, inp, "fp32")
    mm = st["minmax"]
    span = mm[..., 1].amax(-1) - mm[..., 0].amin(-1)
    offs = g["combined_shifts"].float() * span[..., None]
    for pos, proj in ((st["qpos"], st["qproj"]), (st["kpos"], st["kproj"])):
        keys = proj + offs
        want = torch.sort(keys, dim=-1, stable=True).indices
        assert torch.equal(pos.long(), want)
        assert torch.equal(torch.sort(pos.long(), -1).values, torch.arange(n, device=pos.device).expand_as(pos))


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed16"])
@pytest.mark.parametrize("name", ALL)
def test_block_attention_with_reference_permutations(name, precision, gpu_device):
    """The reference's own q/k permutations injected: output must equal the REFERENCE's golden output."""
    inp, fx = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(gpu_device)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(gpu_device)
    st = _staged(g, inp, precision, qp, kp)
    ref = torch.from_numpy(fx["out"])
    out = st["out"].cpu()
    if precision == "fp32":
        # stated tolerance atol 1e-5 / rtol 1e-4 (G3: atol 1e-3) on EVERY element (measured worst: 0.44x of it)
        atol = ATOL.get(name, 1e-5)
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=atol)
    else:
        # bf16 rounds q^/k^ to 8 bits (logit error grows with |q^|); mixed16 keeps 11 bits there, what is left
        # is the bf16 rounding of the weights, v and the stored numerators (2^-9 relative each, unbiased)
        assert _rows_scaled_ok(out, ref, REL16[precision]) >= 0.99
        worst = ((out - ref).abs().amax(-1) / (ref.abs().amax(-1) + 1e-3)).max().item()
        print(f"worst row-scaled error {name} {precision}: {worst:.3e}")
        assert _rows_scaled_ok(out, ref, REL16_ALL_ROWS[name][precision]) == 1.0
        # tight against the oracle's model of the 16-bit path (rounded tiles and weights, fp32 accumulate)
        orc = _oracle(inp, q_positions=qp.long().cpu(), k_positions=kp.long().cpu(), keep=False, **_model_kw(precision))
        # (rtol = 2 bf16 ulps: a last-bit fp32 difference can flip the rounding of a stored bf16 numerator)
        assert _rows_ok(out, orc["out"], atol=5e-3, rtol=8e-3) >= BF16_MODEL_ROWS.get(name, 0.995)
    d = inp["q"].shape[1] // inp["alpha"].shape[0]
    wide = ops.unpack_part(st["part"])
    assert float(wide[..., d + 1:].abs().max()) == 0.0
    if precision != "fp32":
        assert st["part"].shape[-1] == 16 and float(st["part"][..., 13:].abs().max()) == 0.0
    if precision == "fp32" and "denom_rows" in fx:
        rows = torch.from_numpy(fx["rows"].astype(np.int64))
        den = wide[..., d].permute(0, 2, 1).cpu()[..., rows]
        torch.testing.assert_close(den, torch.from_numpy(fx["denom_rows"]), rtol=2e-4, atol=ATOL.get(name, 1e-5) * 10)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed16"])
@pytest.mark.parametrize("name", ALL)
def test_forward_end_to_end_vs_oracle(name, precision, gpu_device):
    """Whole operator (one C call, own radix sort) against the oracle (stable sort): tie-aware row criterion."""
    inp, fx = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    out = _forward(g, inp, precision).cpu()
    orc = _oracle(inp, keep=False, **_model_kw(precision))
    atol = ATOL.get(name, 1e-5) if precision == "fp32" else 5e-3
    rtol = 1e-4 if precision == "fp32" else 8e-3
    assert _rows_ok(out, orc["out"], atol, rtol) >= (0.995 if precision == "fp32" else BF16_MODEL_ROWS.get(name, 0.995))
    # and against the reference itself (unstable argsort there): only tie-induced rows may differ
    ref = torch.from_numpy(fx["out"])
    if precision == "fp32":
        assert _rows_ok(out, ref, atol, rtol) >= 0.98
    else:
        assert _rows_scaled_ok(out, ref, REL16[precision]) >= 0.97
    staged = _staged(g, inp, precision)["out"].cpu()
    assert torch.equal(staged, out)  # hept_forward == the stage entry points chained


def test_reduce_tables_and_partial_forward(gpu_device):
    """Table sharding on one GPU: sum of per-slice partials == all tables; Tl == 1 writes acc directly."""
    inp, _ = cases.load_case("g6_block100")
    g = _gpu(inp, gpu_device)
    n, h, d, e, t = _dims(inp)
    kw = dict(block_size=inp["block_size"], w_per_dist=inp["w_per_dist"], precision="fp32")
    args = (g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"])
    full = ops.forward_partial(*args, t0=0, tl=t, **kw)
    pieces = [ops.forward_partial(*args, t0=i, tl=1, **kw) for i in range(t)]
    torch.testing.assert_close(sum(pieces), full, rtol=1e-6, atol=1e-7)
    # bf16 tiles: packed 64-B partial rows are widened by reduce_tables
    kwb = dict(kw, precision="bf16")
    fullb = ops.forward_partial(*args, t0=0, tl=t, **kwb)
    piecesb = [ops.forward_partial(*args, t0=i, tl=1, **kwb) for i in range(t)]
    torch.testing.assert_close(sum(piecesb), fullb, rtol=1e-6, atol=1e-7)
    outb = ops.combine_out(fullb, d, g["out_weight"], g["out_bias"])
    torch.testing.assert_close(outb, _forward(g, inp, "bf16"), rtol=1e-5, atol=1e-6)
    two = ops.forward_partial(*args, t0=0, tl=2, **kw) + pieces[2]
    torch.testing.assert_close(two, full, rtol=1e-6, atol=1e-7)
    out = ops.combine_out(full, d, g["out_weight"], g["out_bias"])
    torch.testing.assert_close(out, _forward(g, inp, "fp32"), rtol=1e-6, atol=1e-7)
    # point-sliced finishing (reduce-scatter mode) == full finishing
    lo = ops.combine_out(full, d, g["out_weight"], g["out_bias"], 0, 1000)
    hi = ops.combine_out(full, d, g["out_weight"], g["out_bias"], 1000, n - 1000)
    assert torch.equal(torch.cat([lo, hi]), out)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed16"])
def test_tracking_60k_full_size(precision, gpu_device):
    """BASELINE config 3 at full size: golden sampled rows + size-independent properties."""
    inp, fx = cases.load_case("g5_track60k")
    g = _gpu(inp, gpu_device)
    n, h, d, e, t = _dims(inp)
    st = _staged(g, inp, precision)
    out = st["out"]
    # (1) sortedness / permutation / stability of all 2*T*H segments
    mm = st["minmax"]
    span = mm[..., 1].amax(-1) - mm[..., 0].amin(-1)
    offs = g["combined_shifts"].float() * span[..., None]
    for pos, proj in ((st["qpos"], st["qproj"]), (st["kpos"], st["kproj"])):
        keys = proj + offs
        sk = torch.gather(keys, -1, pos.long())
        assert bool((sk[..., 1:] >= sk[..., :-1]).all())
        tie = sk[..., 1:] == sk[..., :-1]
        assert bool((pos[..., 1:][tie] > pos[..., :-1][tie]).all())
        assert torch.equal(torch.sort(pos.long(), -1).values, torch.arange(n, device=pos.device).expand_as(pos))
    # (2) hashes of the sampled rows against the reference
    rows = torch.from_numpy(fx["rows"].astype(np.int64))
    qh_ref = torch.from_numpy(fx["q_hashed_rows"])
    torch.testing.assert_close(st["qproj"].cpu()[..., rows], qh_ref, rtol=0, atol=4e-6 * float(qh_ref.abs().max()))
    # (3) sampled output rows against the reference (unstable sort there -> tie-aware)
    ref = torch.from_numpy(fx["out_rows"])
    if precision == "fp32":
        assert _rows_ok(out.cpu()[rows], ref, 1e-5, 1e-4) >= 0.97
    else:
        assert _rows_scaled_ok(out.cpu()[rows], ref, REL16[precision]) >= 0.96
    # (4) every output row is a convex combination of values pushed through out_linear: linear in v
    g2 = dict(g)
    g2["v"] = g["v"] * 2.0
    out2 = _forward(g2, inp, precision)
    bias = g["out_bias"]
    torch.testing.assert_close(out2 - bias, 2.0 * (out - bias), rtol=1e-4 if precision == "fp32" else 2e-3, atol=1e-5)
    # (5) every row against the full oracle: test_bench_workloads_all_rows_vs_oracle; denominators must be positive
    assert bool((ops.unpack_part(st["part"])[..., d] > 0).all())


def test_edge_cases(gpu_device):
    dev = gpu_device
    from hept_amd.synthetic import make_inputs

    def run(inp, precision="fp32"):
        g = _gpu(inp, dev)
        o = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                        g["out_weight"], g["out_bias"], block_size=inp["block_size"], w_per_dist=10, precision=precision)
        r = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                       inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=inp["block_size"], w_per_dist=10,
                       keep=False)["out"]
        return o.cpu(), r

    # one block only, one table
    inp = make_inputs([64], block_size=64, n_hashes=1, seed=1, cluster_size=4)
    inp["block_size"] = 64
    o, r = run(inp)
    torch.testing.assert_close(o, r, rtol=1e-4, atol=1e-5)
    # 8 tables (HEPT_MAX_TABLES), tiny blocks (B=32 -> one MFMA tile), B=96, B=160, B=224
    for b, t in ((32, 8), (96, 2), (160, 3), (224, 1), (256, 2)):
        inp = make_inputs([b * 5 - 7, b * 3 + 1], block_size=b, n_hashes=t, seed=b, cluster_size=6)
        inp["block_size"] = b
        o, r = run(inp)
        assert _rows_ok(o, r, 1e-5) >= 0.99, (b, t)
    # all AND codes equal (one giant bucket) and all points identical (every key ties: stable = identity order)
    inp = make_inputs([512], block_size=128, n_hashes=2, seed=9)
    inp["block_size"] = 128
    inp["combined_shifts"] = torch.zeros_like(inp["combined_shifts"])
    o, r = run(inp)
    torch.testing.assert_close(o, r, rtol=1e-4, atol=1e-5)
    for key in ("q", "k", "v", "coords"):
        inp[key] = inp[key][:1].expand_as(inp[key]).contiguous()
    o, r = run(inp)
    torch.testing.assert_close(o, r, rtol=1e-4, atol=1e-5)
    # ragged input is an error, as in the reference (EinopsError there)
    g = _gpu(inp, dev)
    with pytest.raises(ValueError, match="multiple of block_size"):
        ops.forward(g["q"][:500], g["k"][:500], g["v"][:500], g["coords"][:500], g["combined_shifts"][..., :500].contiguous(),
                    g["w_rpe_weight"], g["alpha"], g["out_weight"], g["out_bias"], block_size=128, w_per_dist=10)
    with pytest.raises(RuntimeError, match="HEPT_ERR_SHAPE"):
        ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                    g["out_weight"], g["out_bias"], block_size=512, w_per_dist=10)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_v_rows_written_by_the_bucket_sort_launch(precision, gpu_device):
    """Clouds longer than the one-workgroup sort: the v half of the kvhat rows is written by rider workgroups of the
    bucket-sort launch (csrc/sort_tables.hip: RowsJob) and the row builder runs its q and k roles only.  Shapes that
    stress the riders' tiling: a point count that is not a multiple of their 8-point tile, another head count (the
    generic row builder), head dimensions whose 16-bit rows end inside a 16-byte piece (D = 20) or early (D = 8, 16)."""
    from hept_amd.synthetic import make_inputs

    for sizes, b, heads, d, c in (([6300], 100, 8, 24, 6), ([3100, 3300], 100, 4, 16, 4), ([6500], 100, 8, 20, 6),
                                  ([6400], 64, 16, 8, 4), ([7000], 40, 3, 12, 2)):
        inp = make_inputs(sizes, block_size=b, n_hashes=2, num_heads=heads, h_dim=d, coords_dim=c, seed=b + d,
                          cluster_size=6)
        inp["block_size"] = b
        g = _gpu(inp, gpu_device)
        got = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                          g["out_weight"], g["out_bias"], block_size=b, w_per_dist=10, precision=precision).cpu()
        kw = dict(tile_dtype=torch.bfloat16) if precision == "bf16" else {}
        want = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                          inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=b, w_per_dist=10, keep=False,
                          **kw)["out"]
        assert bool(torch.isfinite(got).all()), (sizes, heads, d)
        if precision == "fp32":
            assert _rows_ok(got, want, 1e-5) >= 0.995, (sizes, heads, d)
        else:
            assert _rows_ok(got, want, atol=5e-3, rtol=8e-3) >= 0.995, (sizes, heads, d)


def test_random_shapes_against_the_oracle(gpu_device):
    """tests/op_stress.py: random block sizes (8..256, mostly not multiples of 32), 1..8 tables, every supported
    (head_dim, coords_dim) pair, 1..3 clouds; fp32 and bf16 tiles against the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tests", "op_stress.py"), "18"], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.parametrize("name", ALL)
def test_split_bf16_products_match_the_f32_mfma_kernel(name, gpu_device):
    """precision="fp32" runs the tile products as split-bf16 MFMAs (6 bf16 products per f32 product); the native
    v_mfma_f32_32x32x2_f32 kernel (precision="fp32_mfma") is the in-library ground truth for it: same permutations,
    same f32 rows in, partial rows equal to f32 round-off."""
    inp, _ = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    n, h, d, e, t = _dims(inp)
    st = _staged(g, inp, "fp32")
    ref = ops.block_attn(st["qhat"], st["kvhat"], st["qpos"], st["kpos"], d, inp["block_size"], f32_mfma=True)
    got = st["part"]
    assert got.dtype == torch.float32 and got.shape == ref.shape
    err = (got - ref).abs()
    tol = ATOL.get(name, 1e-5) + 1e-4 * ref.abs()
    assert float((err <= tol).float().mean()) >= 0.995
    assert float((got[..., d + 1:]).abs().max()) == 0.0
    # P enters P.V in TWO bf16 pieces (16 significand bits) since round 4 -- pinned per ROW against the
    # exact f32 kernel on the same rows and blocks: every row of what the combine forms (table-summed numerators over the
    # table-summed denominator, per head) within SPLIT_ROW_X of the fp32 tolerance, dominant-key rows (pileup,
    # checkpoint clouds) included; rows whose denominators are at the 1e-20 floor in every table are 0/0-like in both
    num_g, den_g = got[..., :d].double().sum(0), got[..., d].double().sum(0).unsqueeze(-1)
    num_r, den_r = ref[..., :d].double().sum(0), ref[..., d].double().sum(0).unsqueeze(-1)
    live = den_r.squeeze(-1) > 1e-10
    row_g, row_r = (num_g / den_g)[live], (num_r / den_r)[live]
    row_x = float(((row_g - row_r).abs() / (ATOL.get(name, 1e-5) + 1e-4 * row_r.abs())).max())
    print(f"{name}: per-head rows, split-bf16 (P in two pieces) vs the f32 MFMA kernel: worst element {row_x:.3f} x the tolerance "
          f"({int(live.sum())} of {live.numel()} rows above the denominator floor)")
    assert row_x <= SPLIT_ROW_X
    # the whole operator through the other kernel: both f32 modes agree on >= 99.5 % of rows
    a = _forward(g, inp, "fp32").cpu()
    b = _forward(g, inp, "fp32_mfma").cpu()
    assert _rows_ok(a, b, ATOL.get(name, 1e-5), 1e-4) >= 0.995


@pytest.mark.parametrize("n", [6016, 16384])
def test_sort_is_exact_when_the_sampled_code_maximum_misses_the_largest_codes(n, gpu_device):
    """Round 5: the row builder takes the largest AND code from every 8th tile of 8 points only -- it merely scales the
    sort's bucket ids.  Adversarial case: every SAMPLED point has code 0 and all the others a large code, so the bound is
    far too small and 7/8 of the keys land beyond it (they share the last bucket id: one bucket far over its region's
    capacity at 16 384 points, the one-launch sort at 6 016).  The permutation must still be the exact stable sort."""
    g_ = torch.Generator().manual_seed(23)
    h, d, c, t = 8, 24, 6, 3
    q, k, v = (torch.randn(n, h * d, generator=g_) for _ in range(3))
    coords = torch.randn(n, c, generator=g_)
    alpha = torch.randn(h, d + c, t, generator=g_)
    sqrt_w = torch.rand(h, c, generator=g_) + 0.5
    tile = torch.arange(n) // 8
    codes = torch.where(tile % 8 == 0, torch.zeros(n, dtype=torch.int64), 900 + torch.arange(n) % 37)
    codes = codes.expand(t, h, n).contiguous()
    dev = gpu_device
    ph = ops.prep_hash(q.to(dev), k.to(dev), v.to(dev), coords.to(dev), sqrt_w.to(dev), alpha.to(dev), codes.to(dev), "fp32")
    mm = ph["minmax"]
    assert float(mm[..., 2].max()) == 0.0          # the sample saw code 0 only
    qpos, kpos = ops.sort_tables(ph["qproj"], ph["kproj"], codes.to(dev), mm)
    span = mm[..., 1].amax(-1) - mm[..., 0].amin(-1)
    offs = codes.to(dev).float() * span[..., None]
    for pos, proj in ((qpos, ph["qproj"]), (kpos, ph["kproj"])):
        assert torch.equal(pos.long(), torch.sort(proj + offs, dim=-1, stable=True).indices)


def _almost_sorted(keys, pos, tol):
    """Largest amount by which a key, taken in the order `pos`, lies below an earlier one (0 = sorted)."""
    ks = torch.gather(keys, -1, pos)
    return float((torch.cummax(ks, dim=-1).values - ks).max())


@pytest.mark.parametrize("workload", ["tracking-60k", "pileup-8clouds"])
def test_full_size_every_row_with_the_gpu_permutations_injected(workload, gpu_device):
    """Attribution of every end-to-end mismatch at full size (the exact inputs bench.py times).  (a) With the GPU's OWN
    q/k permutations injected, the oracle reproduces EVERY output row to the fp32 tolerance -- so whatever differs
    end to end differs because of the permutation.  (b) The GPU's permutation differs from the oracle's stable sort
    only where the oracle's keys are tied to within the round-off of the hash (fp32 summation order of the
    30-term projection, <= 4e-6 of the hash scale on either side) plus the rounding of the key addition itself:
    taken in the GPU's order, the oracle's keys are sorted up to that tolerance."""
    from hept_amd.synthetic import WORKLOADS, workload_inputs

    inp = workload_inputs(workload, seed=0)
    inp["block_size"], inp["w_per_dist"] = WORKLOADS[workload]["block_size"], 10
    g = _gpu(inp, gpu_device)
    st = _staged(g, inp, "fp32")
    qp, kp = st["qpos"].long().cpu(), st["kpos"].long().cpu()
    orc = _oracle(inp, q_positions=qp, k_positions=kp, keep=False)
    out = st["out"].cpu()
    assert _rows_ok(out, orc["out"], 1e-5, 1e-4) >= 0.999
    torch.testing.assert_close(out, orc["out"], rtol=HARD_X * 1e-4, atol=HARD_X * 1e-5)      # every element
    own = _oracle(inp, keep=True)
    hash_scale = float(own["q_hashed"].abs().max())
    b, n = inp["block_size"], out.shape[0]
    touched = torch.zeros(n, dtype=torch.bool)   # queries whose block (in any table / head) gained or lost a point
    q_block = None
    for pos, keys, theirs in ((qp, own["q_keys"], own["q_positions"]), (kp, own["k_keys"], own["k_positions"])):
        tol = 8e-6 * hash_scale + 4 * 2.0 ** -23 * float(keys.abs().max())
        assert _almost_sorted(keys, pos, tol) <= tol
        frac = (pos != theirs).float().mean().item()
        print(f"{workload}: {frac:.2e} of the sorted positions differ from the oracle's stable sort")
        assert frac <= 1e-3                      # measured: 1.5e-5 (tracking-60k), 1.4e-6 (pileup)
        rank_gpu = torch.empty_like(pos).scatter_(-1, pos, torch.arange(n).expand_as(pos))
        rank_orc = torch.empty_like(theirs).scatter_(-1, theirs, torch.arange(n).expand_as(theirs))
        changed = (rank_gpu // b) != (rank_orc // b)            # (T, H, N): the point sits in another block
        if q_block is None:
            q_block = rank_gpu // b
            touched |= changed.any(0).any(0)                    # a query that moved to another block itself
        else:                                                   # a key moved: every query of both blocks is affected
            cnt = torch.zeros(pos.shape[0], pos.shape[1], n // b, dtype=torch.int32)
            cnt.scatter_add_(-1, rank_gpu // b, changed.int())
            cnt.scatter_add_(-1, rank_orc // b, changed.int())
            touched |= (cnt > 0).gather(-1, q_block).any(0).any(0)
    # attribution: end to end (the GPU's own sort against the oracle's stable sort) every row that is off by more than
    # the every-element bound is a row whose block membership differs between the two permutations
    off = ~(((out - own["out"]).abs() <= HARD_X * (1e-5 + 1e-4 * own["out"].abs())).all(-1))
    print(f"{workload}: {int(off.sum())} rows off end to end, {int(touched.sum())} rows touched by a moved point")
    assert not bool((off & ~touched).any())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("workload", ["tracking-60k", "pileup-8clouds", "tracking-60k-t8"])
def test_bench_workloads_all_rows_vs_oracle(workload, precision, gpu_device):
    """The exact inputs bench.py times (hept_amd.synthetic.workload_inputs, seed 0), full size, EVERY output row against
    the oracle (a few seconds of CPU at 60k points): tie-aware row criterion as in the end-to-end test.
    tracking-60k-t8 = BASELINE config 4's operator (n_hashes = 8) with all eight tables on this GPU."""
    from hept_amd.synthetic import WORKLOADS, workload_inputs

    inp = workload_inputs(workload, seed=0)
    b = WORKLOADS[workload]["block_size"]
    g = _gpu(inp, gpu_device)
    out = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                      g["out_weight"], g["out_bias"], block_size=b, w_per_dist=10, precision=precision).cpu()
    orc = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                     inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=b, w_per_dist=10, keep=False,
                     **_model_kw(precision))["out"]
    assert bool(torch.isfinite(out).all())
    if precision == "fp32":
        assert _rows_ok(out, orc, 1e-5, 1e-4) >= 0.995
    else:
        assert _rows_ok(out, orc, atol=5e-3, rtol=8e-3) >= 0.995


def test_full_size_with_the_references_own_block_size(gpu_device):
    """block_size = 100, the value of the reference's yaml (src/configs/tracking/tracking_trans_hept.yaml:12), at the
    benchmark's 60 000 points: three full 32-row tiles and one with four rows per block (the ragged kernel variants).
    fp32 tiles: with the GPU's own permutations injected the oracle reproduces EVERY row; 16-bit tiles: the every-row
    bound of the 16-bit modes."""
    from hept_amd.synthetic import make_inputs

    inp = make_inputs([60000], block_size=100, n_hashes=3, seed=0)
    inp["block_size"], inp["w_per_dist"] = 100, 10
    assert inp["q"].shape[0] == 60000
    g = _gpu(inp, gpu_device)
    st = _staged(g, inp, "fp32")
    qp, kp = st["qpos"].long().cpu(), st["kpos"].long().cpu()
    orc = _oracle(inp, q_positions=qp, k_positions=kp, keep=False)
    out = st["out"].cpu()
    assert _rows_ok(out, orc["out"], 1e-5, 1e-4) >= 0.999
    torch.testing.assert_close(out, orc["out"], rtol=HARD_X * 1e-4, atol=HARD_X * 1e-5)    # every element (measured 1.22x)
    assert torch.equal(_forward(g, inp, "fp32").cpu(), out)          # the one-call operator = the staged kernels
    for prec in ("bf16", "mixed16"):
        s16 = _staged(g, inp, prec, qpos=st["qpos"], kpos=st["kpos"])["out"].cpu()
        assert _rows_scaled_ok(s16, orc["out"], REL16[prec]) >= 0.96       # as test_tracking_60k_full_size
        assert _rows_scaled_ok(s16, orc["out"], REL16_ALL_ROWS_FULL[prec]) == 1.0


# G7: the shipped checkpoint's layer-0 weights on UN-RESCALED N(0,1) coordinates.  sqrt_w reaches
# 5.8e3, |q^|^2 ~ 1e8: almost every weight underflows, denominators sit at the 1e-20 floor for many rows, and a
# surviving logit is the difference of 1e8-sized fp32 terms -- the REFERENCE's own output is rounding noise there (an
# error of +-6 in the logit), so no implementation can be held to it element by element.  What is asserted:
# every precision stays finite; fp32 with the reference's permutations reproduces the reference where it is
# well defined -- rows whose reference denominators are all 0 (+1e-20) in every table come out exactly as the
# reference's (bias only), and the well-conditioned majority of rows agree at the G3 tolerance.
SPLIT_ROW_X = 6.0   # measured 0.11-0.16 on the dominant-key cases (pileup, block 100, checkpoint), 2.4 / 3.6 on the default-init
                    # cases g1 / g2, whose worst rows sit just above the denominator floor (total weight ~1e-9: f32 round-off of the logits)
G7_MIN_ROWS_AT_G3_TOL = 0.7   # measured 0.80 (printed below); see DESIGN.md section 4
# against the float64 evaluation of the reference (fixture out_fp64): the fp32 reference itself keeps 81.9 % of its rows
# inside the G3 tolerance (median row error 1.6e-4, worst 0.133); the HIP fp32 output has to do as well, up to this slack
G7_FP64_ROWS_SLACK = 0.02
G7_FP64_FACTOR = 1.5
G7_FP64_FAR_ROWS = 10     # measured 8 of 6016 (the reference: 0); precision="fp32_diff" below: 0
G7_FP64_MEAN_X = 16.0     # mean row error of the fp32 mode over the reference's: measured 13.8 (the far rows carry it)


def test_unrescaled_checkpoint_fp32_is_as_close_to_float64_as_the_reference(gpu_device):
    """Pins the G7 claim (round 5): on the shipped layer-0 scales with raw coordinates the fp32 REFERENCE is rounding
    noise -- so the yardstick is the reference's arithmetic evaluated in float64 on the same blocks (fixture field
    ``out_fp64``, generated by the real reference's stage functions on double inputs with its fp32 permutations).
    The HIP fp32 output (same permutations injected) has to be as close to that as the reference's own fp32 output is:
    as many rows inside the G3 tolerance, no larger typical or worst row error, and no row far off where the
    reference is close."""
    inp, fx = cases.load_case("g7_ckpt_rawcoords")
    g = _gpu(inp, gpu_device)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(gpu_device)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(gpu_device)
    got = _staged(g, inp, "fp32", qp, kp)["out"].cpu().double()
    ref32, ref64 = torch.from_numpy(fx["out"]).double(), torch.from_numpy(fx["out_fp64"])
    tol = ATOL["g3_ckpt6k"] + 1e-4 * ref64.abs()
    e_hip, e_ref = (got - ref64).abs(), (ref32 - ref64).abs()
    rows_hip, rows_ref = float((e_hip <= tol).all(1).float().mean()), float((e_ref <= tol).all(1).float().mean())
    r_hip, r_ref = e_hip.amax(1), e_ref.amax(1)          # worst element of every row
    print(f"g7 vs float64: rows within the G3 tolerance: HIP {rows_hip:.4f}, reference fp32 {rows_ref:.4f}; "
          f"median row error HIP {float(r_hip.median()):.3e} / ref {float(r_ref.median()):.3e}; "
          f"mean {float(r_hip.mean()):.3e} / {float(r_ref.mean()):.3e}; max {float(r_hip.max()):.3e} / {float(r_ref.max()):.3e}; "
          f"rows where HIP is worse than 4x the reference's error and outside the tolerance: "
          f"{int(((r_hip > 4 * r_ref) & ~(e_hip <= tol).all(1)).sum())}")
    assert rows_hip >= rows_ref - G7_FP64_ROWS_SLACK
    assert float(r_hip.median()) <= G7_FP64_FACTOR * float(r_ref.median())
    # ... except for a handful of rows: a row whose only weight is its own key (true logit ~ -0.1, computed as a
    # difference of 3e8-sized terms) is lost when the rounding noise of that difference (sigma ~ 20 in any fp32
    # evaluation, the reference's included) happens to push the logit below -46 = log(1e-20), the denominator floor
    # (example/hept.py:14) -- the output then falls back to the bias.  Measured: 8 of 6016 rows off by more than 0.5
    # (the native f32 MFMA kernel: 4; the reference's own fp32 run: 0 on this seed); asserted as a count, not hidden in
    # a mean
    far = int((r_hip > 0.5).sum())
    print(f"g7 vs float64: rows off by more than 0.5: HIP {far}, reference fp32 {int((r_ref > 0.5).sum())}")
    assert far <= G7_FP64_FAR_ROWS
    # the MEAN is asserted too (round 6): it is 13.8x the reference's, all of it those far rows -- which is why
    # precision="fp32_diff" exists (next test) and is the mode DESIGN.md section 4 names for such inputs
    assert float(r_hip.mean()) <= G7_FP64_MEAN_X * float(r_ref.mean())


@pytest.mark.parametrize("kernel", ["split", "mfma"])
def test_unrescaled_checkpoint_difference_form_is_closer_to_float64_than_the_reference(kernel, gpu_device, monkeypatch):
    """precision="fp32_diff" (round 6): the coordinate part of every logit as -(q^_c - k^_c)^2 / 2 from the stored values
    instead of q^.k^ - |q^|^2/2 - |k^|^2/2 (example/hept.py:8-12) -- the same operator without the cancellation of 3e8-sized
    terms.  On the shipped layer-0 scales with raw coordinates it has to beat the reference's OWN fp32 evaluation against
    the float64 yardstick on every count: more rows inside the G3 tolerance, smaller median, mean and worst row error,
    and no row far off."""
    # two kernels carry the mode: the split-bf16 kernel (D = 24: features through six-term bf16 products) and the f32-MFMA
    # kernel (any D; HEPT_DIFF_MFMA=1 selects it at D = 24 as well)
    monkeypatch.setenv("HEPT_DIFF_MFMA", "1" if kernel == "mfma" else "0")
    inp, fx = cases.load_case("g7_ckpt_rawcoords")
    g = _gpu(inp, gpu_device)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(gpu_device)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(gpu_device)
    got = _staged(g, inp, "fp32_diff", qp, kp)["out"].cpu().double()
    ref32, ref64 = torch.from_numpy(fx["out"]).double(), torch.from_numpy(fx["out_fp64"])
    tol = ATOL["g3_ckpt6k"] + 1e-4 * ref64.abs()
    e_hip, e_ref = (got - ref64).abs(), (ref32 - ref64).abs()
    rows_hip, rows_ref = float((e_hip <= tol).all(1).float().mean()), float((e_ref <= tol).all(1).float().mean())
    r_hip, r_ref = e_hip.amax(1), e_ref.amax(1)
    print(f"g7 fp32_diff ({kernel}) vs float64: rows within the G3 tolerance {rows_hip:.4f} (reference fp32 {rows_ref:.4f}); median "
          f"{float(r_hip.median()):.3e} / {float(r_ref.median()):.3e}; mean {float(r_hip.mean()):.3e} / {float(r_ref.mean()):.3e}; "
          f"max {float(r_hip.max()):.3e} / {float(r_ref.max()):.3e}; rows off by more than 0.5: {int((r_hip > 0.5).sum())}")
    assert rows_hip >= rows_ref
    assert float(r_hip.median()) <= float(r_ref.median())
    assert float(r_hip.mean()) <= float(r_ref.mean())
    assert float(r_hip.max()) <= float(r_ref.max())
    assert int((r_hip > 0.5).sum()) == 0


@pytest.mark.parametrize("kernel", ["split", "mfma"])
@pytest.mark.parametrize("name", ["g1_rand512", "g3_ckpt6k", "g4_pileup", "g6_block100"])
def test_difference_form_matches_the_reference_where_it_is_well_conditioned(name, kernel, gpu_device, monkeypatch):
    """The same mode on the ordinary golden cases (reference permutations injected): every element of the reference's
    output at the fp32 tolerance -- the difference form changes how a logit is summed, not what it is."""
    monkeypatch.setenv("HEPT_DIFF_MFMA", "1" if kernel == "mfma" else "0")
    inp, fx = cases.load_case(name)
    g = _gpu(inp, gpu_device)
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(gpu_device)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(gpu_device)
    got = _staged(g, inp, "fp32_diff", qp, kp)["out"].cpu()
    torch.testing.assert_close(got, torch.from_numpy(fx["out"]), rtol=1e-4, atol=ATOL.get(name, 1e-5))
    # ... and through the module (its own sort): the one-call operator runs the same kernel
    out = _forward(g, inp, "fp32_diff")
    assert bool(torch.isfinite(out).all()) and _rows_ok(out.cpu(), torch.from_numpy(fx["out"]), ATOL.get(name, 1e-5), 1e-4) >= 0.99


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed16"])
def test_unrescaled_checkpoint_coordinates_stay_finite(precision, gpu_device):
    inp, fx = cases.load_case("g7_ckpt_rawcoords")
    g = _gpu(inp, gpu_device)
    out = _forward(g, inp, precision)
    assert bool(torch.isfinite(out).all())
    qp = torch.from_numpy(fx["q_positions"].astype(np.int32)).to(gpu_device)
    kp = torch.from_numpy(fx["k_positions"].astype(np.int32)).to(gpu_device)
    st = _staged(g, inp, precision, qp, kp)
    got, ref = st["out"].cpu(), torch.from_numpy(fx["out"])
    assert bool(torch.isfinite(got).all())
    # the output is a convex combination of value rows pushed through out_linear: bounded by the values themselves
    bound = float(inp["v"].abs().max()) * float(inp["out_weight"].abs().sum(dim=1).max()) + float(inp["out_bias"].abs().max())
    assert float(got.abs().max()) <= bound * 1.001
    frac = _rows_ok(got, ref, atol=ATOL["g3_ckpt6k"], rtol=1e-4)
    print(f"g7 {precision}: rows within the G3 tolerance of the reference: {frac:.4f}; "
          f"median |err| {float((got - ref).abs().median()):.3e}, max |err| {float((got - ref).abs().max()):.3e}, "
          f"|ref| mean {float(ref.abs().mean()):.3f}")
    if precision == "fp32":
        # rows the reference leaves at 0 / 1e-20 in every table and head: bias only, bit for bit
        dead = torch.from_numpy(fx["out"] == inp["out_bias"].numpy()[None, :]).all(dim=1)
        assert torch.equal(got[dead], ref[dead])
        assert frac >= G7_MIN_ROWS_AT_G3_TOL


@pytest.mark.parametrize("workload", ["tracking-60k", "pileup-8clouds"])
@pytest.mark.parametrize("precision", ["fp32", "fp32_diff", "bf16"])
def test_full_size_properties_linearity_in_the_values_and_determinism(workload, precision, gpu_device):
    """Size-independent properties at the benchmark's full sizes (no oracle involved).  For fixed q, k the operator is
    LINEAR in v before the bias: numerators are sums of weight x value, the denominators do not see v (example/hept.py:13-17),
    so f(a v1 + b v2) - bias = a (f(v1) - bias) + b (f(v2) - bias) with the same blocks (the hashes do not see v either).
    And it is deterministic: two calls are torch.equal (no float atomics, the sort is exact, the region sort's arrival order
    does not reach the output)."""
    from hept_amd.synthetic import workload_inputs, WORKLOADS

    inp = workload_inputs(workload, seed=0)
    bs = WORKLOADS[workload]["block_size"]
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    gen = torch.Generator(device="cpu").manual_seed(17)
    v2 = torch.randn(g["v"].shape, generator=gen).to(gpu_device)

    def f(v):
        return ops.forward(g["q"], g["k"], v, g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                           g["out_weight"], g["out_bias"], block_size=bs, w_per_dist=10, precision=precision)

    a, b = 0.75, -1.5
    o1, o2, o12 = f(g["v"]), f(v2), f(a * g["v"] + b * v2)
    assert torch.equal(o1, f(g["v"]))                                   # determinism
    bias = g["out_bias"]
    lhs, rhs = o12 - bias, a * (o1 - bias) + b * (o2 - bias)
    scale = float(rhs.abs().max())
    err = float((lhs - rhs).abs().max()) / scale
    # f32 tiles: the round-off of three weighted means; 16-bit tiles: values and stored numerators are rounded to bf16
    assert err <= (2e-5 if precision != "bf16" else 2e-2), err
