"""Drop-in nn.Module behaviour on the GPU (-m gpu): reference-shaped checkpoint, kwargs, dtypes, errors."""
import numpy as np
import pytest
import torch

import cases
import hept_oracle as ho
from hept_amd import HEPTAttention, ops

pytestmark = pytest.mark.gpu


def _module(inp, dev, **extra):
    h, e, t = inp["alpha"].shape
    d = inp["q"].shape[1] // h
    # the reference constructs it with the whole model-config dict (extra keys must be tolerated)
    m = HEPTAttention(e, h_dim=d, num_heads=h, block_size=inp["block_size"], n_hashes=t, num_w_per_dist=10,
                      n_layers=4, num_regions=150, pe_type="none", **extra)
    sd = {"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]}
    if extra.get("variant") == "src":
        sd["e2lsh.beta"] = torch.zeros(1, t)
    m.load_state_dict(sd, strict=True)
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0])
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    return m.to(dev).eval(), w_rpe.to(dev)


@pytest.mark.parametrize("name", ["g1_rand512", "g3_ckpt6k"])
def test_module_forward_matches_reference_golden(name, gpu_device):
    inp, fx = cases.load_case(name)
    m, w_rpe = _module(inp, gpu_device)
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    with torch.no_grad():
        out = m(g["q"], g["k"], g["v"], pe=g["coords"], w_rpe=w_rpe, coords=g["coords"],
                combined_shifts=g["combined_shifts"])
    assert out.shape == (inp["q"].shape[0], 24) and out.dtype == torch.float32
    ref = torch.from_numpy(fx["out"])
    err = (out.cpu() - ref).abs()
    atol = 1e-3 if name == "g3_ckpt6k" else 1e-5
    assert ((err <= atol + 1e-4 * ref.abs()).all(-1)).float().mean() >= 0.98
    direct = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"],
                         g["out_weight"], g["out_bias"], block_size=inp["block_size"], w_per_dist=10)
    assert torch.equal(out, direct)


def test_module_bf16_and_errors(gpu_device):
    inp, fx = cases.load_case("g6_block100")
    m, w_rpe = _module(inp, gpu_device, precision="bf16")
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        out = m(g["q"], g["k"], g["v"], **kw)
        again = m(g["q"], g["k"], g["v"], **kw)
    assert torch.equal(out, again)  # deterministic run to run (no atomics on the data path)
    ref = torch.from_numpy(fx["out"])
    err = (out.cpu() - ref).abs().amax(-1)
    assert (err <= 2.5e-2 * (ref.abs().amax(-1) + 1e-3)).float().mean() >= 0.97
    # a 16-bit module trains through the f32-tile kernels (the reference trains through the same operator,
    # example/trainer.py:11-22): same values as the fp32 module, gradients flow
    q = g["q"].clone().requires_grad_(True)
    out_t = m(q, g["k"], g["v"], **kw)
    m32, _ = _module(inp, gpu_device)
    with torch.no_grad():
        out32 = m32(g["q"], g["k"], g["v"], **kw)
    torch.testing.assert_close(out_t, out32, rtol=1e-5, atol=1e-6)
    out_t.sum().backward()
    assert q.grad is not None and bool(q.grad.abs().sum() > 0)
    # model.eval(); model(x) WITHOUT no_grad and without input gradients: served by the inference path (one warning)
    with pytest.warns(UserWarning, match="runs the inference path"):
        out_e = m(g["q"], g["k"], g["v"], **kw)
    assert torch.equal(out_e, out) and not out_e.requires_grad
    with torch.no_grad(), pytest.raises(ValueError, match="multiple of block_size"):
        m(g["q"][:150], g["k"][:150], g["v"][:150], w_rpe=w_rpe, coords=g["coords"][:150],
          combined_shifts=g["combined_shifts"][..., :150])
    with pytest.raises(ValueError):
        HEPTAttention(30, h_dim=24, num_heads=8, block_size=64, n_hashes=2, num_w_per_dist=10, precision="fp8")


def test_more_tables_than_one_chunk(gpu_device):
    """n_hashes = 10 > HEPT_MAX_TABLES: the entry points walk the tables in chunks (the reference takes any n_hashes,
    example/hept.py:37-41)."""
    from hept_amd.synthetic import make_inputs

    inp = make_inputs([1500, 700], block_size=128, n_hashes=10, seed=3, cluster_size=8)
    inp["block_size"] = 128
    ref = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                     inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=128, w_per_dist=10, keep=False)["out"]
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    for precision, atol, rtol in (("fp32", 1e-5, 1e-4), ("bf16", 2e-2, 2e-2)):
        m, w_rpe = _module(inp, gpu_device, precision=precision)
        with torch.no_grad():
            out = m(g["q"], g["k"], g["v"], w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"]).cpu()
        ok = ((out - ref).abs() <= atol + rtol * ref.abs()).all(-1).float().mean().item()
        assert ok >= 0.99, (precision, ok)
        # tables [0, 8) + [8, 10) as two partial calls, summed: the same rows as the one-call chunked walk
        if precision == "fp32":
            args = (g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], g["w_rpe_weight"], g["alpha"])
            a = ops.forward_partial(*args, block_size=128, w_per_dist=10, t0=0, tl=8)
            b = ops.forward_partial(*args, block_size=128, w_per_dist=10, t0=8, tl=2)
            whole = ops.forward_partial(*args, block_size=128, w_per_dist=10, t0=0, tl=10)
            torch.testing.assert_close(whole, a + b, rtol=2e-5, atol=1e-30)  # ((t0..t7) + t8) + t9 vs (t0..t7) + (t8 + t9)
            # ... and the TRAINING path takes more than one chunk of tables too (same values, gradients against the
            # oracle's autograd)
            m.train()
            qg = g["q"].clone().requires_grad_(True)
            out_t = m(qg, g["k"], g["v"], w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
            ok = ((out_t.detach().cpu() - ref).abs() <= atol + rtol * ref.abs()).all(-1).float().mean().item()
            assert ok >= 0.99, ok
            gup = torch.randn(out_t.shape, generator=torch.Generator().manual_seed(5))
            (out_t * gup.to(gpu_device)).sum().backward()
            qc = inp["q"].clone().requires_grad_(True)
            o = ho.forward(qc, inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                           inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=128, w_per_dist=10, keep=False,
                           grad=True)["out"]
            (o * gup).sum().backward()
            scale = float(qc.grad.abs().max())
            assert float(((qg.grad.cpu() - qc.grad).abs() <= 2e-4 * scale).float().mean()) >= 0.99
            m.eval()


@pytest.mark.parametrize("variant", ["example", "src"])
def test_module_under_torch_compile_is_one_graph(variant, gpu_device):
    """torch.compile(model) as in example/example.ipynb:161: with the operator registered through torch.library
    Dynamo captures the module call as ONE graph (fullgraph=True raises on any graph break) and the compiled
    module returns what the eager one returns.  backend='aot_eager': capture + functionalisation, no codegen."""
    import torch._dynamo

    if variant == "example":
        inp, _ = cases.load_case("g1_rand512")
        extra = dict(combined_shifts=inp["combined_shifts"].to(gpu_device))
    else:
        inp, _ = cases.load_case_src("s1_src1000")
        extra = dict(raw_size=inp["raw_size"], regions_h=inp["regions_h"].to(gpu_device),
                     region_indices=[inp["eta_idx"].to(gpu_device), inp["phi_idx"].to(gpu_device)])
    m, w_rpe = _module(inp, gpu_device, **({"variant": "src"} if variant == "src" else {}))
    q, k, v, coords = (inp[x].to(gpu_device) for x in ("q", "k", "v", "coords"))

    def run(mod):
        with torch.no_grad():
            return mod(q, k, v, w_rpe=w_rpe, coords=coords, **extra)

    eager = run(m)
    torch._dynamo.reset()
    compiled = torch.compile(m, backend="aot_eager", fullgraph=True)
    out = run(compiled)
    assert torch.equal(out, eager)


def test_forward_is_hip_graph_capturable(gpu_device):
    """The C call launches on the caller's stream without allocating or synchronising, so the whole forward can be
    captured into a HIP graph (torch.cuda.CUDAGraph) and replayed: same result, ~7 launches collapse into one."""
    inp, _ = cases.load_case("g1_rand512")
    m, w_rpe = _module(inp, gpu_device, precision="bf16")
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        eager = m(g["q"], g["k"], g["v"], **kw)          # warm-up: library, workspace, kernel attributes
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(g["q"], g["k"], g["v"], **kw)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = m(g["q"], g["k"], g["v"], **kw)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        g["v"].mul_(2.0)                                   # replay reads the live input buffers
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(out, m(g["q"], g["k"], g["v"], **kw), rtol=0, atol=0)


def test_module_on_a_device_that_is_not_current(gpu_device):
    """The C ABI launches on the thread's current HIP device; the front end makes the tensors' device current for the
    call.  Needs two visible GPUs (the driver's 8-GPU box); on a 1-GPU box the guard is exercised with the same device."""
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", 1 if n_dev > 1 else 0)
    inp, fx = cases.load_case("g4_pileup")   # block 256 with f32 tiles: a kernel that needs the > 64 KiB LDS attribute
    m, w_rpe = _module(inp, dev)
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
    torch.cuda.set_device(0)
    with torch.no_grad():
        out = m(g["q"], g["k"], g["v"], w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    assert out.device == dev and torch.cuda.current_device() == 0
    ref = torch.from_numpy(fx["out"])
    assert ((out.cpu() - ref).abs() <= 1e-5 + 1e-4 * ref.abs()).all(-1).float().mean() >= 0.98


def test_reserve_sizes_the_workspace_before_the_first_call(gpu_device):
    """reserve(): the first forward finds its workspace in place (no device allocation inside the call), and a smaller
    cloud keeps using it; the values are those of a module that grew its workspace on demand"""
    inp, _ = cases.load_case("g1_rand512")
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    m, w_rpe = _module(inp, gpu_device)
    plain, _ = _module(inp, gpu_device)
    assert m._workspace is None
    m.reserve(4 * inp["q"].shape[0], g["coords"].shape[1], "cuda")
    ptr, size = m._workspace.data_ptr(), m._workspace.numel()
    assert m._workspace.device == g["q"].device
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    with torch.no_grad():
        out = m(g["q"], g["k"], g["v"], **kw)
        ref = plain(g["q"], g["k"], g["v"], **kw)
    assert (m._workspace.data_ptr(), m._workspace.numel()) == (ptr, size)
    assert plain._workspace.numel() < size
    assert torch.equal(out, ref)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_in_place_update_of_w_rpe_through_data_is_seen(precision, gpu_device):
    """``w.weight.data.mul_()/copy_()`` (EMA swaps, hand-rolled optimisers) changes a parameter WITHOUT bumping its
    ``_version`` and without moving it: the reference recomputes ``sqrt_w`` from the weight on every forward
    (example/hept.py:21-28), and so does the row builder here -- both forwards must match the oracle."""
    inp, _ = cases.load_case("g1_rand512")
    m, w_rpe = _module(inp, gpu_device, precision=precision)
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
    model = dict(tile_dtype=torch.bfloat16) if precision == "bf16" else {}

    def oracle(weight):
        return ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], weight, inp["alpha"],
                          inp["out_weight"], inp["out_bias"], block_size=inp["block_size"], w_per_dist=10, keep=False,
                          **model)["out"]

    def rows_ok(out, ref):
        atol, rtol = (1e-5, 1e-4) if precision == "fp32" else (5e-3, 8e-3)
        return ((out.cpu() - ref).abs() <= atol + rtol * ref.abs()).all(-1).float().mean().item()

    with torch.no_grad():
        first = m(g["q"], g["k"], g["v"], **kw)
    assert rows_ok(first, oracle(inp["w_rpe_weight"])) >= 0.99
    version, ptr = w_rpe.weight._version, w_rpe.weight.data_ptr()
    w_rpe.weight.data.mul_(1.5)                       # in place, through .data
    w_rpe.weight.data[:, :10].add_(0.01)
    assert w_rpe.weight._version == version and w_rpe.weight.data_ptr() == ptr   # nothing a cache key could see
    new_w = w_rpe.weight.detach().cpu().clone()
    with torch.no_grad():
        second = m(g["q"], g["k"], g["v"], **kw)
    ref2 = oracle(new_w)
    assert rows_ok(second, ref2) >= 0.99
    assert rows_ok(first, ref2) < 0.5                 # the update does change the answer: a stale scale would fail
    # the stage entry point with an explicit sqrt_w agrees bit for bit with the in-kernel weight math
    sw = ops.rpe_scale(w_rpe.weight.detach(), m.num_heads, m.dim_per_head, 10)
    staged = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], sw, g["alpha"], g["out_weight"],
                         g["out_bias"], block_size=inp["block_size"], w_per_dist=0, precision=precision)
    assert torch.equal(staged, second)


def test_out_linear_weight_at_an_odd_storage_offset(gpu_device):
    """the D = 24 combine reads out_linear.weight and the partial rows as 16-byte pieces.  A contiguous
    view at an odd storage offset is legal for the module (ops makes an aligned copy); a raw pointer that is not 16-byte
    aligned is refused by the C ABI with HEPT_ERR_ARG instead of faulting inside the kernel."""
    from hept_amd import _lib
    inp, _ = cases.load_case("g1_rand512")
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    sw = ops.rpe_scale(g["w_rpe_weight"], 8, 24, 10)
    want = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], sw, g["alpha"], g["out_weight"],
                       g["out_bias"], block_size=inp["block_size"], w_per_dist=0)
    store = torch.zeros(g["out_weight"].numel() + 1, device=gpu_device)
    odd = store[1:].view_as(g["out_weight"])
    odd.copy_(g["out_weight"])
    assert odd.is_contiguous() and odd.data_ptr() % 16 == 4
    got = ops.forward(g["q"], g["k"], g["v"], g["coords"], g["combined_shifts"], sw, g["alpha"], odd, g["out_bias"],
                      block_size=inp["block_size"], w_per_dist=0)
    assert torch.equal(got, want)
    lib = _lib.load()
    n, h, d = g["q"].shape[0], 8, 24
    part = torch.zeros(1, n, h, 32, device=gpu_device)
    out = torch.empty(n, d, device=gpu_device)
    rc = lib.hept_combine_out(part.data_ptr(), 0, 1, n, h, d, 0, n, odd.data_ptr(), None, out.data_ptr(), None)
    assert rc == 3   # HEPT_ERR_ARG
    rc = lib.hept_combine_out(part.data_ptr(), 0, 1, n, h, d, 0, n, g["out_weight"].data_ptr(), None, out.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0


def test_combine_without_the_staged_rows_gives_the_same_bits(gpu_device):
    """the staged combine (rows through LDS, > 64 KB of dynamic LDS for f32 rows with the feed-forward
    epilogue on a short cloud) has a lane-by-lane fallback (taken when the device refuses the LDS size; forced here with
    HEPT_NO_STAGED_COMBINE=1): both must give the same bits -- the fused Attn block on tracking-6k, fp32 and bf16."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "from hept_amd import ops\n"
        "from hept_amd.synthetic import workload_inputs\n"
        "inp = workload_inputs('tracking-6k', seed=3)\n"
        "dev = torch.device('cuda:0')\n"
        "g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}\n"
        "gen = torch.Generator().manual_seed(5)\n"
        "r = lambda *s: (torch.randn(*s, generator=gen) * 0.2).to(dev)\n"
        "x = torch.randn(g['q'].shape[0], 24, generator=gen).to(dev)\n"
        "p = {'norm1.weight': r(24) + 1, 'norm1.bias': r(24), 'w_q.weight': r(192, 24), 'w_k.weight': r(192, 24),\n"
        "     'w_v.weight': r(192, 24), 'w_rpe.weight': g['w_rpe_weight'], 'attn.e2lsh.alpha': g['alpha'],\n"
        "     'attn.out_linear.weight': g['out_weight'], 'attn.out_linear.bias': g['out_bias'], 'norm2.weight': r(24) + 1,\n"
        "     'norm2.bias': r(24), 'ff.0.weight': r(24, 24), 'ff.0.bias': r(24), 'ff.2.weight': r(24, 24), 'ff.2.bias': r(24)}\n"
        "for prec in ('fp32', 'bf16'):\n"
        "    y = ops.attn_block_forward(x, g['coords'], g['combined_shifts'], p, num_heads=8, block_size=128, w_per_dist=10,\n"
        "                               precision=prec)\n"
        "    print(prec, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest(), bool(torch.isfinite(y).all()))\n"
    ) % root
    outs = []
    for switch in ("0", "1"):
        env = dict(os.environ, HEPT_NO_STAGED_COMBINE=switch)
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert run.returncode == 0, run.stderr[-2000:]
        outs.append([ln for ln in run.stdout.splitlines() if ln.startswith(("fp32", "bf16"))])
    assert len(outs[0]) == 2 and outs[0] == outs[1] and all(ln.endswith("True") for ln in outs[0]), outs


def test_two_streams_share_one_module_safely(gpu_device):
    """One module owns one workspace.  Round 6 enforces what round 5 only documented: a forward issued on another stream
    than the one before it waits for that stream (one event wait), so alternating streams cannot overwrite rows an earlier
    forward is still reading -- every result equals the single-stream one."""
    from hept_amd import HEPTAttention
    from hept_amd.synthetic import make_inputs

    inp = make_inputs([3000, 1500], block_size=128, n_hashes=3, seed=3, cluster_size=8)
    g = {k: v.to(gpu_device) for k, v in inp.items() if torch.is_tensor(v)}
    m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=128, n_hashes=3, num_w_per_dist=10, precision="bf16")
    m.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"], "e2lsh.alpha": inp["alpha"]})
    m = m.to(gpu_device).eval()
    w_rpe = torch.nn.Linear(50, 192).to(gpu_device)
    with torch.no_grad():
        w_rpe.weight.copy_(g["w_rpe_weight"])
        kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])
        ref = [m(g["q"] * (1.0 + 0.1 * i), g["k"], g["v"], **kw).clone() for i in range(6)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=gpu_device) for _ in range(2)]
        outs = []
        for i in range(6):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(m(g["q"] * (1.0 + 0.1 * i), g["k"], g["v"], **kw))
        torch.cuda.synchronize()
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)


def test_the_launch_order_of_the_block_attention_does_not_change_the_output(gpu_device):
    """Round 6 runs the tables of a block side by side (csrc/block_attn.hip: HeadRange::tl) instead of table-major.  The map
    only decides WHEN a (table, head, block) runs: the output of the round-5 map (HEPT_ATTN_TABLE_MAJOR=1, read once per
    process -- hence the child process) is bit-identical, in every precision."""
    import hashlib
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "from hept_amd import ops\n"
        "from hept_amd.synthetic import make_inputs\n"
        "inp = make_inputs([9000, 4000], block_size=128, n_hashes=3, seed=4, cluster_size=8)\n"
        "g = {k: v.cuda() for k, v in inp.items() if torch.is_tensor(v)}\n"
        "for p in ('fp32', 'bf16', 'mixed16'):\n"
        "    o = ops.forward(g['q'], g['k'], g['v'], g['coords'], g['combined_shifts'], g['w_rpe_weight'], g['alpha'],\n"
        "                    g['out_weight'], g['out_bias'], block_size=128, w_per_dist=10, precision=p)\n"
        "    print(p, hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())\n" % root)
    outs = []
    for extra in ({}, {"HEPT_ATTN_TABLE_MAJOR": "1"}):
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **extra), timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        outs.append([ln for ln in run.stdout.splitlines() if ln.split() and ln.split()[0] in ("fp32", "bf16", "mixed16")])
    assert len(outs[0]) == 3 and outs[0] == outs[1], outs
