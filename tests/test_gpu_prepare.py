"""prepare_input on the GPU (-m gpu): HIP kernels vs the host mirror (which make_golden.py
pinned on the reference's own prepare_input) and vs the reference's stored padding."""
import numpy as np
import pytest
import torch

import cases
from hept_amd import ops
from hept_amd.prep import prepare_input, prepare_input_hip

pytestmark = pytest.mark.gpu


def _cpu_prepare(inp, block_size):
    helper = {"block_size": block_size, "num_heads": cases.NUM_HEADS, "regions": inp["regions"]}
    n_raw = inp["n_raw"]
    return prepare_input(torch.arange(n_raw), inp["coords_raw"], inp["batch"], helper)


@pytest.mark.parametrize("name", ["g2_example4k", "g3_ckpt6k", "g4_pileup", "g5_track60k", "g6_block100"])
def test_hip_prepare_equals_host_mirror(name, gpu_device):
    """Both use stable sorts, so everything is bit-exact: pad_seq, mask, AND codes, padded coords."""
    inp, fx = cases.load_case(name)
    b = cases.CASES[name]["block_size"]
    pad_cpu, kw_cpu, mask_cpu = _cpu_prepare(inp, b)
    helper = {"block_size": b, "num_heads": cases.NUM_HEADS, "regions": inp["regions"].to(gpu_device)}
    x = torch.arange(inp["n_raw"], device=gpu_device)
    pad_gpu, kw_gpu, mask_gpu = prepare_input_hip(x, inp["coords_raw"].to(gpu_device), inp["batch"].to(gpu_device), helper)
    assert torch.equal(pad_gpu.cpu(), pad_cpu)
    assert torch.equal(mask_gpu.cpu(), mask_cpu)
    assert torch.equal(kw_gpu["combined_shifts"].cpu(), kw_cpu["combined_shifts"])
    assert torch.equal(kw_gpu["coords"].cpu(), kw_cpu["coords"])
    # against the reference: identical on real points up to the stored tie patches, same pad code multiset
    assert torch.equal(pad_gpu.cpu()[mask_cpu], torch.arange(inp["n_raw"]))
    ref_pad = torch.from_numpy(fx["pad_seq"].astype(np.int64))
    raw_codes = kw_gpu["combined_shifts"].cpu()[0, 0][mask_cpu]
    assert torch.equal(torch.sort(raw_codes[pad_gpu.cpu()[~mask_cpu]]).values,
                       torch.sort(raw_codes[ref_pad[~mask_cpu]]).values)


def test_segmented_argsort_is_stable_and_handles_padding(gpu_device):
    g = torch.Generator().manual_seed(5)
    keys = torch.randn(7, 5000, generator=g)
    keys[:, ::3] = keys[:, 1::3][:, : keys[:, ::3].shape[1]]  # many exact ties
    keys[2, 4000:] = float("inf")                               # padding sorts last, in index order
    keys[5] = 1.25                                              # all equal: identity permutation
    keys[6, :100] = -0.0
    keys[6, 100:200] = 0.0                                      # -0.0 == +0.0
    pos = ops.segmented_argsort(keys.to(gpu_device)).long().cpu()
    want = torch.sort(keys, dim=-1, stable=True).indices
    assert torch.equal(pos, want)
    one = torch.randn(1, 77, generator=g)
    assert torch.equal(ops.segmented_argsort(one.to(gpu_device)).long().cpu(), torch.sort(one, dim=-1, stable=True).indices)


def test_segmented_argsort_skewed_and_long_segments(gpu_device):
    """The sort's slow paths: a bucket larger than the LDS tile (streamed through global scratch), one id group
    holding most of a segment, and segments long enough for the large-tile kernel."""
    g = torch.Generator().manual_seed(6)
    keys = torch.randn(5, 20000, generator=g)
    keys[0, :18000] = keys[0, :18000] * 1e-4 + 3.0          # 90 % of the keys inside 1/1000 of the range
    keys[1, 2000:] = 7.5                                     # one id group of 18 000 equal keys (ties by index)
    keys[2] = torch.randint(0, 50, (20000,), generator=g).float()            # 50 distinct values, huge tie groups
    keys[3, ::2] = float("inf")                              # half the segment is padding
    keys[3, 1] = float("-inf")
    keys[4] = torch.linspace(1.0, 1.0 + 1e-3, 20000).flip(0)  # descending, all keys nearly equal
    pos = ops.segmented_argsort(keys.to(gpu_device)).long().cpu()
    assert torch.equal(pos, torch.sort(keys, dim=-1, stable=True).indices)
    long_keys = torch.randn(2, 700000, generator=g)          # average bucket 2734 pairs: the large-tile kernel
    long_keys[1] = (long_keys[1] * 4).round() / 4           # quantised: ~100 distinct values
    pos = ops.segmented_argsort(long_keys.to(gpu_device)).long().cpu()
    assert torch.equal(pos, torch.sort(long_keys, dim=-1, stable=True).indices)


def test_degenerate_segments_stay_fast(gpu_device):
    """All keys equal (zero-initialised features hash to 0 everywhere) used to cost O(N^2) compares per segment --
    minutes at tracking-60k.  Oversize buckets are now regrouped by a monotone id of the (key, index) pair."""
    import time

    n = 60032
    keys = torch.zeros(48, n)
    keys[1] = 3.25
    keys[2, n // 2:] = 1.0                       # two piles
    keys[3] = torch.arange(n).float() // 5000     # 13 piles of 5000
    keys[4, -200:] = float("inf")                # src-variant padding behind a pile
    dev_keys = keys.to(gpu_device)
    ops.segmented_argsort(dev_keys[:1, :4096].contiguous())  # warm-up (library load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pos = ops.segmented_argsort(dev_keys)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.equal(pos.long().cpu(), torch.sort(keys, dim=-1, stable=True).indices)
    assert dt < 0.5, f"degenerate sort took {dt:.3f} s"


def test_prepare_then_attention_pipeline(gpu_device):
    """prepare_input (HIP) -> HEPTAttention (HIP) end to end on GPU tensors only, against the CPU pipeline."""
    import hept_oracle as ho
    inp, _ = cases.load_case("g4_pileup")
    b = 256
    dev = gpu_device
    helper = {"block_size": b, "num_heads": 8, "regions": inp["regions"].to(dev)}
    g = torch.Generator().manual_seed(1)
    n_raw = inp["n_raw"]
    feats = torch.randn(n_raw, 192 * 3, generator=g)
    x_pad, kw, mask = prepare_input(feats.to(dev), inp["coords_raw"].to(dev), inp["batch"].to(dev), helper)
    q, k, v = x_pad[:, :192].contiguous(), x_pad[:, 192:384].contiguous(), x_pad[:, 384:].contiguous()
    out = ops.forward(q, k, v, kw["coords"], kw["combined_shifts"], inp["w_rpe_weight"].to(dev), inp["alpha"].to(dev),
                      inp["out_weight"].to(dev), inp["out_bias"].to(dev), block_size=b, w_per_dist=10)
    pad_cpu, kw_cpu, _ = _cpu_prepare(inp, b)
    fc = feats[pad_cpu]
    ref = ho.forward(fc[:, :192], fc[:, 192:384], fc[:, 384:], kw_cpu["coords"], kw_cpu["combined_shifts"],
                     inp["w_rpe_weight"], inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=b, w_per_dist=10,
                     keep=False)["out"]
    err = (out.cpu() - ref).abs()
    assert ((err <= 1e-5 + 1e-4 * ref.abs()).all(-1)).float().mean() >= 0.995
    assert out[mask].shape[0] == n_raw


def test_segmented_argsort_randomised(gpu_device):
    """Random sizes / segment counts / key distributions (tools/sort_stress.py runs 400 of these): exact every time."""
    import subprocess
    import sys
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "sort_stress.py"), "60"], capture_output=True,
                         text=True)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]


def test_segmented_argsort_ragged(gpu_device):
    """Ragged segments: only the first lens[s] keys of a row take part (whatever lies behind them, NaN included);
    pos[s, :lens[s]] equals the stable sort of that prefix.  Covers the one-workgroup path (short rows), the
    bucket path (long rows) and rows of length 0 / 1."""
    g = torch.Generator().manual_seed(31)
    for s_, l_ in [(6, 700), (5, 9000), (3, 70000), (40, 33)]:
        keys = torch.randn(s_, l_, generator=g)
        if l_ == 9000:
            keys = keys.round()  # heavy ties: stability matters
        lens = torch.randint(0, l_ + 1, (s_,), generator=g, dtype=torch.int32)
        lens[0], lens[-1] = l_, min(1, l_)
        if s_ > 2:
            lens[1] = 0
        poisoned = keys.clone()
        for s in range(s_):
            poisoned[s, int(lens[s]):] = float("nan") if s % 2 else -1e30
        pos = ops.segmented_argsort(poisoned.to(gpu_device), lens.to(gpu_device)).long().cpu()
        for s in range(s_):
            n = int(lens[s])
            assert torch.equal(pos[s, :n], torch.sort(keys[s, :n], stable=True).indices), (s_, l_, s, n)


@pytest.mark.parametrize("n", [9000, 60032, 131072])
def test_segmented_argsort_region_overflow_pool(gpu_device, n):
    """Round 5 layout: every (segment, bucket) owns a region of fixed capacity (2 048 pairs up to 65 536 keys); runs
    that do not fit go to the segment's overflow pool and the bucket collects them from there.  Three hot buckets per
    segment far above the capacity (pairs of SEVERAL buckets interleave in the pool, several chunks overflow into it),
    one of them a pile of equal keys, next to ordinary buckets; plus a segment that is one pile."""
    g = torch.Generator().manual_seed(17)
    keys = torch.rand(4, n, generator=g)
    hot = n // 5
    for s_ in range(3):
        perm = torch.randperm(n, generator=g)
        keys[s_, perm[:hot]] = 0.25 + torch.rand(hot, generator=g) / 300            # one bucket, spread
        keys[s_, perm[hot:2 * hot]] = 0.5 + torch.rand(hot, generator=g) / 2000      # one bucket, its lower part
        keys[s_, perm[2 * hot:3 * hot]] = 0.75                                       # one id group
        keys[s_, 0], keys[s_, 1] = 0.0, 1.0
    keys[3] = -2.5
    pos = ops.segmented_argsort(keys.to(gpu_device)).long().cpu()
    assert torch.equal(pos, torch.sort(keys, dim=-1, stable=True).indices)


@pytest.mark.parametrize("spread", ["whole_bucket", "lower_half", "one_bin"])
def test_segmented_argsort_oversize_bucket_paths(gpu_device, spread):
    """60 000 uniform keys + 1 500 extra keys inside one of the 256 top-level buckets: the bucket holds ~1 730 pairs,
    more than the 1 024-pair LDS tile.  Spread over the whole bucket it takes the two-pass LDS path, packed into its
    lower half (or a single id bin) the halves do not fit and the splitter streaming path takes over.  Exact either way."""
    g = torch.Generator().manual_seed(5)
    n = 60000
    keys = torch.rand(3, n + 1500, generator=g)
    width = {"whole_bucket": 1 / 256, "lower_half": 1 / 600, "one_bin": 1e-7}[spread]
    keys[:, n:] = 100 / 256 + torch.rand(3, 1500, generator=g) * width
    keys[:, 0], keys[:, 1] = 0.0, 1.0  # pin the key range so that the bucket boundaries are where the test expects
    keys = keys[:, torch.randperm(n + 1500, generator=g)]
    pos = ops.segmented_argsort(keys.to(gpu_device)).long().cpu()
    assert torch.equal(pos, torch.sort(keys, dim=-1, stable=True).indices)


def test_region_table_updated_in_place_is_re_read(gpu_device):
    """The 2^24 guard of the pad sort reads the region counts from the tensor on every call: growing them in place
    through ``.data`` (no ``_version`` bump) must trip it on the next call instead of reusing a remembered width."""
    inp, _ = cases.load_case("g2_example4k")
    b = cases.CASES["g2_example4k"]["block_size"]
    regions = inp["regions"].to(gpu_device).clone()
    helper = {"block_size": b, "num_heads": cases.NUM_HEADS, "regions": regions}
    x = torch.arange(inp["n_raw"], device=gpu_device)
    args = (x, inp["coords_raw"].to(gpu_device), inp["batch"].to(gpu_device), helper)
    prepare_input_hip(*args)
    v = regions._version
    regions.data.mul_(4096.0)            # 12 more bits per axis: codes no longer fit an exact fp32 key
    assert regions._version == v
    with pytest.raises(ValueError, match="2\\^24"):
        prepare_input_hip(*args)


@pytest.mark.parametrize("n_clouds,dtype", [(1, torch.int64), (7, torch.int32), (255, torch.int64), (256, torch.int64),
                                            (300, torch.int32)])
def test_cloud_boundaries_from_the_probe_kernel(n_clouds, dtype, gpu_device):
    """``hept_prepare_probe`` (one kernel + a 32-byte host record) sizes the outputs for batches of up to 255 clouds;
    larger batches take the torch path.  Both against the host mirror, with int64 and int32 batch vectors and clouds of
    very different sizes (1 point up to a few hundred)."""
    g = torch.Generator().manual_seed(n_clouds)
    sizes = torch.randint(1, 40, (n_clouds,), generator=g)
    sizes[0] = 1
    sizes[-1] = 333
    batch = torch.repeat_interleave(torch.arange(n_clouds), sizes)
    n_raw = int(sizes.sum())
    coords = torch.randn(n_raw, 4, generator=g)
    regions = torch.tensor([[[2.0] * cases.NUM_HEADS, [3.0] * cases.NUM_HEADS]] * 2)   # (T = 2, 2, H): 2 x 3 regions
    helper = {"block_size": 16, "num_heads": cases.NUM_HEADS, "regions": regions}
    pad_cpu, kw_cpu, mask_cpu = prepare_input(torch.arange(n_raw), coords, batch, helper)
    helper_g = {"block_size": 16, "num_heads": cases.NUM_HEADS, "regions": regions.to(gpu_device)}
    pad_gpu, kw_gpu, mask_gpu = prepare_input_hip(torch.arange(n_raw, device=gpu_device), coords.to(gpu_device),
                                                  batch.to(gpu_device, dtype), helper_g)
    assert torch.equal(pad_gpu.cpu(), pad_cpu) and torch.equal(mask_gpu.cpu(), mask_cpu)
    assert torch.equal(kw_gpu["combined_shifts"].cpu(), kw_cpu["combined_shifts"])
    assert torch.equal(kw_gpu["coords"].cpu(), kw_cpu["coords"])


def test_a_cloud_id_without_points_is_an_error(gpu_device):
    batch = torch.tensor([0] * 20 + [2] * 20)     # id 1 owns nothing
    coords = torch.randn(40, 4)
    helper = {"block_size": 8, "num_heads": cases.NUM_HEADS,
              "regions": torch.full((2, 2, cases.NUM_HEADS), 2.0, device=gpu_device)}
    with pytest.raises(ValueError, match="at least one point"):
        prepare_input_hip(torch.arange(40, device=gpu_device), coords.to(gpu_device), batch.to(gpu_device), helper)
