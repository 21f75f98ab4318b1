"""The C ABI driven by a PyTorch-free C++ host (-m gpu): tests/c_host/hept_host.cpp is compiled with hipcc against
include/hept_hip.h, linked to libhept_hip.so, fed a golden case through a binary file and compared with the
reference's golden output.  This is the drop-in boundary a non-Python maintainer would bind."""
import os
import subprocess

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,precision", [("g1_rand512", 0), ("g6_block100", 0), ("g6_block100", 1)])
def test_cpp_host_runs_the_operator(name, precision, tmp_path, gpu_device):
    src = os.path.join(ROOT, "tests", "c_host", "hept_host.cpp")
    lib_dir = os.path.join(ROOT, "hept_amd", "csrc")
    exe = str(tmp_path / "hept_host")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", src, "-o", exe, f"-L{lib_dir}", "-lhept_hip",
                    f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True)
    inp, fx = cases.load_case(name)
    h, e, t = inp["alpha"].shape
    n = inp["q"].shape[0]
    d, c = inp["q"].shape[1] // h, inp["coords"].shape[1]
    prob = tmp_path / "problem.bin"
    with open(prob, "wb") as f:
        np.asarray([n, h, d, c, inp["w_per_dist"], t, inp["block_size"], precision, 0], dtype=np.int32).tofile(f)
        for key in ("q", "k", "v", "coords", "w_rpe_weight", "alpha", "out_weight", "out_bias"):
            inp[key].contiguous().numpy().astype(np.float32).tofile(f)
        inp["combined_shifts"].contiguous().numpy().astype(np.int64).tofile(f)
    out_path = tmp_path / "out.bin"
    run = subprocess.run([exe, str(prob), str(out_path)], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    out = torch.from_numpy(np.fromfile(out_path, dtype=np.float32).reshape(n, d))
    ref = torch.from_numpy(fx["out"])
    err = (out - ref).abs()
    if precision == 0:  # fp32 tiles: the reference's golden output up to sort ties
        assert float(((err <= 1e-5 + 1e-4 * ref.abs()).all(-1)).float().mean()) >= 0.98
    else:               # bf16 tiles: row-scaled tolerance of the 16-bit mode (tests/test_gpu_parity.py)
        assert float((err.amax(-1) <= 2.5e-2 * (ref.abs().amax(-1) + 1e-3)).float().mean()) >= 0.98


@pytest.mark.parametrize("world,precision", [(1, 0), (2, 0), (3, 1)])
def test_cpp_host_runs_the_sharded_operator(world, precision, tmp_path, gpu_device):
    """tests/c_host/hept_host_sharded.cpp: `world` forked processes, no PyTorch, no RCCL -- hept_comm_create_local, exchange
    buffers swapped as HIP IPC handles over socket pairs, hept_forward_sharded with the one-sided transport -- against
    the single-process hept_forward of the same library (the C host above)."""
    lib_dir = os.path.join(ROOT, "hept_amd", "csrc")
    exes = {}
    for name in ("hept_host", "hept_host_sharded"):
        exes[name] = str(tmp_path / name)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "c_host", name + ".cpp"), "-o",
                        exes[name], f"-L{lib_dir}", "-lhept_hip", f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True)
    inp, _ = cases.load_case("g6_block100")
    h, e, t = inp["alpha"].shape
    n = inp["q"].shape[0]
    d, c = inp["q"].shape[1] // h, inp["coords"].shape[1]
    prob = tmp_path / "problem.bin"
    with open(prob, "wb") as f:
        np.asarray([n, h, d, c, inp["w_per_dist"], t, inp["block_size"], precision, 0], dtype=np.int32).tofile(f)
        for key in ("q", "k", "v", "coords", "w_rpe_weight", "alpha", "out_weight", "out_bias"):
            inp[key].contiguous().numpy().astype(np.float32).tofile(f)
        inp["combined_shifts"].contiguous().numpy().astype(np.int64).tofile(f)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([exes["hept_host"], str(prob), str(tmp_path / "plain.bin")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    run = subprocess.run([exes["hept_host_sharded"], str(prob), str(tmp_path / "sharded.bin"), str(world)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr
    plain = torch.from_numpy(np.fromfile(tmp_path / "plain.bin", dtype=np.float32).reshape(n, d))
    shard = torch.from_numpy(np.fromfile(tmp_path / "sharded.bin", dtype=np.float32).reshape(n, d))
    if precision == 0:
        torch.testing.assert_close(shard, plain, rtol=1e-5, atol=1e-6)
    else:   # packed rows: a rank with two tables rounds its table sum to bf16 once more before it travels
        err = (shard - plain).abs().amax(-1)
        assert bool((err <= 4e-3 * (plain.abs().amax(-1) + 1e-2)).all())
