"""The C-ABI shared library: builds, loads, exports every symbol include/hept_hip.h declares.
No compute call is made here (no GPU in this container); pure host entry points are exercised."""
import ctypes
import os
import re

import pytest

from hept_amd import _lib
from hept_amd.build import LIB_PATH, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hept_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hept_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(lib):
    names = _declared_symbols()
    assert len(names) >= 12
    raw = ctypes.CDLL(LIB_PATH)
    for nm in names:
        assert hasattr(raw, nm), f"{nm} declared in include/hept_hip.h but not exported"
        assert nm in _lib.SIGNATURES, f"{nm} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_constants(lib):
    assert lib.hept_abi_version() == _lib.ABI_VERSION
    text = open(os.path.join(ROOT, "include", "hept_hip.h")).read()
    for macro, val in (("HEPT_ROW", _lib.ROW), ("HEPT_MAX_TABLES", _lib.MAX_TABLES),
                       ("HEPT_MAX_BLOCK", _lib.MAX_BLOCK), ("HEPT_PREP_GRID", _lib.PREP_GRID)):
        assert re.search(rf"#define {macro} {val}\b", text)


def test_shape_checks_are_host_side(lib):
    ok = lib.hept_check_shape
    assert ok(60032, 8, 24, 6, 3, 128) == 0
    assert ok(5120, 8, 24, 4, 3, 256) == 0
    assert ok(2400, 8, 24, 6, 3, 100) == 0
    assert ok(60000, 8, 24, 6, 3, 128) == 1      # N % B != 0 (the reference raises EinopsError)
    assert ok(4096, 4, 24, 6, 3, 128) == 0       # any head count up to 16 ...
    assert ok(4096, 16, 20, 5, 3, 128) == 0      # ... and any head / coordinate dims that fit the 32-column rows
    assert ok(4096, 17, 24, 6, 3, 128) == 1
    assert ok(4096, 8, 28, 2, 3, 128) == 1       # head dim > 27
    assert ok(4096, 8, 26, 6, 3, 128) == 1       # D + C > 30
    assert ok(4096, 8, 24, 6, 9, 128) == 0       # any number of tables (walked in chunks of HEPT_MAX_TABLES)
    assert ok(4096, 8, 24, 6, 0, 128) == 1
    assert ok(4096, 8, 24, 6, 3, 512) == 1       # block too large
    assert ok(4096, 8, 20, 6, 3, 128) == 0


def test_workspace_size_model(lib):
    n, h, t = 60032, 8, 3
    bf = lib.hept_workspace_bytes(n, h, 24, 6, t, 128, _lib.PREC_BF16)
    fp = lib.hept_workspace_bytes(n, h, 24, 6, t, 128, _lib.PREC_F32)
    rows = h * n * 96
    assert fp - bf == pytest.approx(rows * 2, rel=1e-3)
    part = t * n * h * 32 * 4
    assert bf > part + rows * 2 and bf < 2.5 * (part + rows * 2)
    assert lib.hept_sort_workspace_bytes(n, h, t) > 2 * t * h * n * (4 + 4 + 4)


def test_null_pointers_are_rejected_before_any_launch(lib):
    assert lib.hept_rpe_scale(None, 8, 24, 6, 10, None, None) == 3
    assert lib.hept_block_attn(None, None, None, None, 128, 8, 24, 1, 128, 0, None, None) == 3
    assert lib.hept_forward(*([None] * 9), 128, 8, 24, 6, 10, 3, 128, 0, None, 0, None, None) == 3


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_module_refuses_cpu_tensors():
    import torch
    from hept_amd import HEPTAttention

    m = HEPTAttention(30, h_dim=24, num_heads=8, block_size=64, n_hashes=2, num_w_per_dist=10, n_layers=4, pe_type="none")
    assert set(m.state_dict()) == {"out_linear.weight", "out_linear.bias", "e2lsh.alpha"}
    assert tuple(m.e2lsh.alpha.shape) == (8, 30, 2) and tuple(m.out_linear.weight.shape) == (24, 192)
    q = torch.zeros(128, 192)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        m(q, q, q, w_rpe=torch.nn.Linear(50, 192), coords=torch.zeros(128, 6),
          combined_shifts=torch.zeros(2, 8, 128, dtype=torch.int64))


def test_custom_ops_trace_with_fake_tensors():
    """torch.library registration (hept_amd/library.py): under FakeTensorMode the ops run their shape-only kernels,
    so a compiler can capture the operator as one node without touching the GPU library."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from hept_amd import library  # noqa: F401  (registers the ops)

    with FakeTensorMode():
        n = 1024
        f = lambda *s, dt=torch.float32: torch.empty(*s, device="cuda", dtype=dt)  # noqa: E731
        out = torch.ops.hept_amd.forward(f(n, 192), f(n, 192), f(n, 192), f(n, 6), f(3, 8, n, dt=torch.int64),
                                         f(192, 50), f(8, 30, 3), f(24, 192), f(24), 128, 10, "bf16")
        assert tuple(out.shape) == (n, 24) and out.dtype == torch.float32 and out.device.type == "cuda"
        out = torch.ops.hept_amd.forward_src(f(n, 192), f(n, 192), f(n, 192), f(n, 6), f(24, n), f(24, n), f(2, 24),
                                             1000, f(192, 50), f(8, 30, 3), f(24, 192), None, 128, 10, "fp32")
        assert tuple(out.shape) == (n, 24)
    # no CPU kernel is registered: CPU tensors are refused by the dispatcher
    import pytest
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.hept_amd.forward(torch.zeros(128, 192), torch.zeros(128, 192), torch.zeros(128, 192),
                                   torch.zeros(128, 6), torch.zeros(3, 8, 128, dtype=torch.int64), torch.zeros(192, 50),
                                   torch.zeros(8, 30, 3), torch.zeros(24, 192), None, 128, 10, "fp32")


def test_header_is_plain_c():
    """include/hept_hip.h is the drop-in boundary for any FFI: it has to compile as C99 and as C++ on its own."""
    import os
    import subprocess

    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hept_hip.h")
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-Wall", "-Werror", hdr], check=True)
