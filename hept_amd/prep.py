"""Caller-side input preparation for the HEPT operator.

Host-side mirror of the step that runs immediately before
``HEPTAttention.forward`` in the reference model shell: quantile regions in
(eta, phi), the packed AND code ``combined_shifts`` and padding of every cloud
to a multiple of ``block_size``.  Same names, argument meaning and return
values as the reference (``example/transformer.py:10-63``,
``example/hept_utils.py:6-31``); written with torch tensor ops only, so it runs
on whatever device the inputs live on (no per-cloud Python loops over sorts).

Behavioural notes kept from the reference:

* ``batch`` must be sorted (points of one cloud contiguous), as the reference
  assumes when it slices by ``bincount().cumsum()``
  (``example/transformer.py:40-47``).
* Padding slots of cloud ``i`` replicate real points: the first ``pad_i`` points
  of the window of the last ``block_size`` positions of cloud ``i`` in the
  order sorted by table-0/head-0 AND code (``example/transformer.py:24-31``).
  The reference sorts with an unstable ``argsort``; this mirror uses a stable
  sort, so pads are drawn from the same code window but may be different
  members of a tie group.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

__all__ = ["get_regions", "quantile_partition", "bit_shift", "pad_and_unpad", "prepare_input", "prepare_input_hip",
           "prepare_input_src", "prepare_input_src_hip"]


def get_regions(num_regions, num_or_hashes, num_heads, num_and_hashes=2, generator=None) -> torch.Tensor:
    """Random (eta, phi) region counts per (table, head); shape (T, A, H).

    Reference: ``example/hept_utils.py:17-31``.  Each (table, head) pair draws
    ``num_and_hashes`` factors uniformly in ``[2, 2·R^(1/A) - 2]``, rescales them
    so that their product is ``num_regions`` and rounds to thirds.
    """
    lo = 2.0
    hi = 2.0 * num_regions ** (1.0 / num_and_hashes) - lo
    draws = torch.rand(num_or_hashes * num_heads, num_and_hashes, generator=generator)
    factors = draws * (hi - lo) + lo
    scale = (num_regions / factors.prod(dim=1, keepdim=True)) ** (1.0 / num_and_hashes)
    factors = torch.round(scale * factors * 3) / 3
    # rows are ordered (head-major inside a table): "(h c) a -> c a h"
    return factors.reshape(num_heads, num_or_hashes, num_and_hashes).permute(1, 2, 0).contiguous()


def quantile_partition(sorted_indices: torch.Tensor, num_regions: torch.Tensor) -> torch.Tensor:
    """Region id (1-based, float) of every point from its rank; ``example/hept_utils.py:6-14``.

    ``sorted_indices``: (n,) argsort of one coordinate.  ``num_regions``: (R, 1)
    target region counts.  Returns (R, n): ``rank // ceil(n / num_regions) + 1``.
    """
    n = sorted_indices.shape[-1]
    # the reference writes ``n / num_regions`` with a Python int numerator, which torch
    # evaluates as ``num_regions.reciprocal() * n`` (one extra rounding); keep that form
    width = torch.ceil(num_regions.reciprocal() * n)
    rank = torch.empty_like(sorted_indices)
    rank[sorted_indices] = torch.arange(n, device=sorted_indices.device)
    return rank[None] // width + 1


def bit_shift(base: torch.Tensor, shift_idx: torch.Tensor) -> torch.Tensor:
    """Pack ``shift_idx`` above the bits used by ``base`` (row-wise); ``example/transformer.py:10-13``."""
    top = base.max(dim=1, keepdim=True).values
    n_bits = torch.ceil(torch.log2(top + 1)).long()
    return (shift_idx << n_bits) | base


def pad_and_unpad(batch, block_size, region_indices, raw_sizes) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gather index that pads each cloud to a multiple of ``block_size`` and the mask that undoes it.

    Reference: ``example/transformer.py:16-32``.
    """
    dev = batch.device
    padded = ((raw_sizes + block_size - 1) // block_size) * block_size
    n_pad = padded - raw_sizes
    raw_end = raw_sizes.cumsum(0)
    raw_start = raw_end - raw_sizes
    pad_end = padded.cumsum(0)
    pad_start = pad_end - padded
    total = int(pad_end[-1])

    cloud = torch.repeat_interleave(torch.arange(len(raw_sizes), device=dev), padded)
    slot = torch.arange(total, device=dev) - pad_start[cloud]
    is_real = slot < raw_sizes[cloud]

    by_code = torch.sort(region_indices, stable=True).indices
    # pad slot j of cloud i copies the point at sorted position raw_end[i] - block_size + j
    src_sorted = raw_end[cloud] - block_size + (slot - raw_sizes[cloud])
    src_sorted = torch.where(is_real, torch.zeros_like(src_sorted), src_sorted)
    pad_seq = torch.where(is_real, raw_start[cloud] + slot, by_code[src_sorted])
    return pad_seq, is_real


def _rank_within_cloud(values: torch.Tensor, batch: torch.Tensor, raw_start: torch.Tensor) -> torch.Tensor:
    """Rank of each point among the points of its own cloud, ordered by ``values`` (segmented argsort)."""
    by_val = torch.sort(values, stable=True).indices
    by_cloud = torch.sort(batch[by_val], stable=True).indices
    order = by_val[by_cloud]  # grouped by cloud, ascending value inside
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=values.device)
    return rank - raw_start[batch]


_MAX_CLOUDS_ONE_TRIP = 255


def prepare_input_hip(x, coords, batch, helper_params) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """``prepare_input`` on the GPU through the C ABI (``hept_prepare_input``, ``csrc/prepare.hip``).

    Same contract as :func:`prepare_input`.  The only host round trip is the per-cloud size vector
    (needed to size the outputs; the reference synchronises on it as well).
    """
    from . import _lib

    lib = _lib.load()
    regions = helper_params["regions"].to(device=coords.device, dtype=torch.float32).contiguous()
    block_size, num_heads = int(helper_params["block_size"]), int(helper_params["num_heads"])
    n_tables = regions.shape[0]
    if regions.shape[2] != num_heads:
        raise ValueError("regions must have shape (n_hashes, 2, num_heads)")
    coords_c = coords.contiguous().float()
    n_raw, c_dim = coords_c.shape
    dev = coords.device
    # ONE host round trip (needed to size the outputs): the cloud boundaries of up to _MAX_CLOUDS_ONE_TRIP clouds are
    # searched on the device (`batch` is sorted, so they are a searchsorted; boundaries past the last cloud equal
    # n_raw) and travel together with the region maxima; the cloud count is read off the boundaries on the host.  A
    # batch of more clouds pays a second trip for its count.
    # The region maxima: packed codes must stay below 2^24 (they are sorted as exact fp32 keys), and the largest region
    # counts per axis are read from the tensor on EVERY call -- no cache keyed on a version counter that `.data` updates
    # do not bump.  (The concatenation promotes the boundaries to float32 -- exact below 2^24 points.)
    import math

    stream = torch.cuda.current_stream(dev)
    if batch.dtype in (torch.int64, torch.int32) and n_raw < (1 << 30):
        # the usual batch (at most 255 clouds): ONE small kernel finds the cloud boundaries, scans the padded sizes and
        # reads the region maxima; the host waits for its 32-byte record in pinned memory and does no tensor work at all
        # (csrc/prepare.hip: probe_kernel).  Anything it does not resolve takes the torch path below.
        batch_c = batch.contiguous()
        bounds = torch.empty(2 * 257, device=dev, dtype=torch.int32)       # cloud_start | pad_start
        record = torch.zeros(8, dtype=torch.int32, pin_memory=True)
        _lib.check(lib.hept_prepare_probe(batch_c.data_ptr(), 1 if batch.dtype == torch.int64 else 0, n_raw, block_size,
                                          regions.data_ptr(), n_tables, num_heads, bounds.data_ptr(),
                                          bounds[257:].data_ptr(), record.data_ptr(), stream.cuda_stream),
                   "hept_prepare_probe")
        stream.synchronize()
        rec = record.tolist()
        if rec[7] != 0x600DF00D:
            raise RuntimeError("hept_prepare_probe: the host record was not written")
        if not rec[4]:
            n_clouds, n_pad, max_cloud, smallest = rec[0], rec[1], rec[2], rec[3]
            if smallest < 1:
                raise ValueError("every cloud id in [0, batch.max()] must own at least one point")
            reg_hi = torch.tensor(rec[5:7], dtype=torch.int32).view(torch.float32).tolist()
            bits = sum((int(math.ceil(float(v))) + 1).bit_length() for v in reg_hi)
            if (n_clouds << bits) >= (1 << 24):
                raise ValueError("AND codes would exceed 2^24: too many clouds x regions for the fp32-keyed pad sort")
            return _prepare_outputs(lib, x, coords, coords_c, c_dim, bounds[:257], bounds[257:], n_clouds, n_raw,
                                    max_cloud, n_pad, regions, n_tables, num_heads, block_size, stream)
    probe = _MAX_CLOUDS_ONE_TRIP
    wide = n_raw >= (1 << 24)
    while True:
        edges = torch.searchsorted(batch.contiguous(), torch.arange(probe + 1, device=dev, dtype=batch.dtype))
        reg_hi = regions.amax(dim=(0, 2))
        host = (torch.cat([edges.double(), reg_hi.double()]) if wide else torch.cat([edges, reg_hi])).cpu()
        if int(host[probe]) >= n_raw:
            break
        probe = int(batch[-1]) + 1      # more clouds than the probe: ask for the count
    n_clouds = int((host[:probe + 1] < n_raw).sum())
    host = torch.cat([host[:n_clouds + 1], host[probe + 1:]])
    cloud_start = edges[:n_clouds + 1].to(torch.int32)
    sizes = host[:n_clouds + 1].diff().long()
    if int(sizes.min()) < 1:
        raise ValueError("every cloud id in [0, batch.max()] must own at least one point")
    padded = ((sizes + block_size - 1) // block_size) * block_size
    max_cloud, n_pad = int(sizes.max()), int(padded.sum())
    import math

    bits = sum((int(math.ceil(float(host[n_clouds + 1 + a]))) + 1).bit_length() for a in (0, 1))
    if (n_clouds << bits) >= (1 << 24):
        raise ValueError("AND codes would exceed 2^24: too many clouds x regions for the fp32-keyed pad sort")
    pad_start = torch.cat([torch.zeros(1, dtype=torch.int64), padded.cumsum(0)]).to(torch.int32).to(dev, non_blocking=True)
    return _prepare_outputs(lib, x, coords, coords_c, c_dim, cloud_start, pad_start, n_clouds, n_raw, max_cloud, n_pad,
                            regions, n_tables, num_heads, block_size, stream)


def _prepare_outputs(lib, x, coords, coords_c, c_dim, cloud_start, pad_start, n_clouds, n_raw, max_cloud, n_pad, regions,
                     n_tables, num_heads, block_size, stream):
    """The outputs of ``prepare_input`` once their sizes are known: one ``hept_prepare_input`` call and the feature
    gather."""
    from . import _lib

    dev = coords.device
    ws = torch.empty(int(lib.hept_prepare_workspace_bytes(n_raw, n_clouds, max_cloud, n_tables, num_heads)),
                     device=dev, dtype=torch.uint8)
    pad_seq = torch.empty(n_pad, device=dev, dtype=torch.int64)
    unpad = torch.empty(n_pad, device=dev, dtype=torch.bool)   # one byte per slot, written as 0 / 1
    coords_pad = torch.empty(n_pad, c_dim, device=dev, dtype=torch.float32)
    codes_pad = torch.empty(n_tables, num_heads, n_pad, device=dev, dtype=torch.int64)
    _lib.check(lib.hept_prepare_input(coords_c.data_ptr(), c_dim, cloud_start.data_ptr(), pad_start.data_ptr(),
                                      n_clouds, n_raw, max_cloud, n_pad, regions.data_ptr(), n_tables, num_heads,
                                      block_size, ws.data_ptr(), ws.numel(), pad_seq.data_ptr(), unpad.data_ptr(),
                                      coords_pad.data_ptr(), codes_pad.data_ptr(), stream.cuda_stream),
               "hept_prepare_input")
    kwargs = {"combined_shifts": codes_pad, "coords": coords_pad.to(coords.dtype)}
    return x[pad_seq], kwargs, unpad.bool()


def prepare_input(x, coords, batch, helper_params) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """Padded features, ``{"combined_shifts", "coords"}`` and the un-pad mask.

    Reference: ``example/transformer.py:35-63``.  ``helper_params`` carries
    ``block_size``, ``num_heads`` and ``regions`` of shape (T, 2, H).  GPU tensors go through the HIP
    kernels (:func:`prepare_input_hip`); CPU tensors through the torch implementation below, which is
    also the host-side specification the HIP path is tested against.
    """
    if coords.is_cuda:
        return prepare_input_hip(x, coords, batch, helper_params)
    regions = helper_params["regions"]
    block_size, num_heads = helper_params["block_size"], helper_params["num_heads"]
    n_tables = regions.shape[0]
    per_axis = regions.permute(1, 0, 2).reshape(2, n_tables * num_heads)  # "c a h -> a (c h)"
    with torch.no_grad():
        sizes = batch.bincount()
        raw_end = sizes.cumsum(0)
        raw_start = raw_end - sizes
        n_in_cloud = sizes[batch]

        region_ids = []
        for axis in (0, 1):
            rank = _rank_within_cloud(coords[:, axis], batch, raw_start)
            # reciprocal-times-n, as in quantile_partition (see the note there)
            width = torch.ceil(per_axis[axis][:, None].reciprocal() * n_in_cloud[None])
            region_ids.append((rank[None] // width + 1).long())
        codes = bit_shift(region_ids[0], region_ids[1])
        codes = bit_shift(codes, batch[None])
        codes = codes.reshape(n_tables, num_heads, -1)

        pad_seq, unpad_seq = pad_and_unpad(batch, block_size, codes[0, 0], sizes)
        kwargs = {"combined_shifts": codes[..., pad_seq], "coords": coords[pad_seq]}
        return x[pad_seq], kwargs, unpad_seq


def _stable_argsort_rows(keys: torch.Tensor) -> torch.Tensor:
    """Stable ascending argsort of every row; GPU rows go through the library's exact segmented sort."""
    if keys.is_cuda:
        from . import ops

        return ops.segmented_argsort(keys.float().contiguous()).long()
    return torch.sort(keys, dim=-1, stable=True).indices


def prepare_input_src(x, coords, helper_funcs) -> Tuple[torch.Tensor, Dict]:
    """Caller-side preparation of the reference's ``src`` variant; ``src/models/baselines/transformer.py:43-57``.

    One cloud per call (the reference asserts ``batch.max() == 0``).  Pads ``x`` with zero rows and ``coords`` with
    +inf rows up to a multiple of ``block_size``, ranks eta (``coords[:, 0]``) and phi (``coords[:, 1]``) with the
    pads last, turns the ranks into float region ids per (table, head) with ``quantile_partition``, then zeroes the
    pad coordinates.  Returns the padded ``x`` and the kwargs of ``HEPTAttention.forward``: ``raw_size``,
    ``coords``, ``region_indices`` = [eta (T*H, N), phi (T*H, N)] and ``regions_h`` (2, T*H).  Ties are broken by
    ascending index (the reference's ``argsort`` leaves them undefined); runs on the device of its inputs.
    """
    block_size, regions = int(helper_funcs["block_size"]), helper_funcs["regions"]
    if coords.is_cuda:
        return prepare_input_src_hip(x, coords, block_size, regions)
    with torch.no_grad():
        raw_size = x.shape[0]
        n_pad = (-raw_size) % block_size
        x_p = torch.cat([x, x.new_zeros((n_pad,) + tuple(x.shape[1:]))]) if n_pad else x
        coords_p = torch.cat([coords, coords.new_full((n_pad, coords.shape[1]), float("inf"))]) if n_pad \
            else coords.clone()
        order = _stable_argsort_rows(coords_p[:, :2].t().contiguous())
        n_tables, _, num_heads = regions.shape
        regions_h = regions.permute(1, 0, 2).reshape(2, n_tables * num_heads)  # "c a h -> a (c h)"
        region_indices = [quantile_partition(order[axis], regions_h[axis][:, None]) for axis in (0, 1)]
        coords_p[raw_size:] = 0.0
    return x_p, {"raw_size": raw_size, "coords": coords_p, "region_indices": region_indices, "regions_h": regions_h}


def prepare_input_src_hip(x, coords, block_size: int, regions) -> Tuple[torch.Tensor, Dict]:
    """:func:`prepare_input_src` for GPU tensors in ONE C call (``hept_prepare_input_src``): padding of features and
    coordinates, the two stable ranks (padding last), the float region ids of every (table, head) and the zeroing
    of the padded coordinates.  Bit-exact against the torch implementation above (tests/test_gpu_src.py)."""
    from . import _lib
    from .ops import _stream

    lib = _lib.load()
    raw_size = x.shape[0]
    n = raw_size + (-raw_size) % block_size
    n_tables, _, num_heads = regions.shape
    dev = coords.device
    regions_f = regions.to(device=dev, dtype=torch.float32).contiguous()
    coords_f = coords.float().contiguous()
    feat = int(x[0].numel()) if raw_size else 0
    # the HIP padding kernel is outside autograd: features that carry a gradient are padded by torch.cat, as in the
    # reference (src/models/baselines/transformer.py:43-57), so that the gradient reaches the caller's tensor
    x_is_f32 = (x.is_cuda and x.dtype == torch.float32 and feat > 0
                and not (x.requires_grad and torch.is_grad_enabled()))
    x_f = x.reshape(raw_size, feat).contiguous() if x_is_f32 else None
    x_pad = torch.empty((n, feat), device=dev, dtype=torch.float32) if x_is_f32 else None
    coords_pad = torch.empty((n, coords.shape[1]), device=dev, dtype=torch.float32)
    eta = torch.empty((n_tables * num_heads, n), device=dev, dtype=torch.float32)
    phi = torch.empty_like(eta)
    ws = torch.empty(int(lib.hept_prepare_src_workspace_bytes(n)), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _lib.check(lib.hept_prepare_input_src(x_f.data_ptr() if x_is_f32 else None, feat, coords_f.data_ptr(),
                                              coords.shape[1], raw_size, n, regions_f.data_ptr(), n_tables, num_heads,
                                              ws.data_ptr(), ws.numel(), x_pad.data_ptr() if x_is_f32 else None,
                                              coords_pad.data_ptr(), eta.data_ptr(), phi.data_ptr(), _stream(coords)),
                   "hept_prepare_input_src")
    if x_is_f32:
        x_p = x_pad.reshape((n,) + tuple(x.shape[1:]))
    else:  # other dtypes / devices of x: pad with torch (the reference pads whatever it is given)
        x_p = torch.cat([x, x.new_zeros((n - raw_size,) + tuple(x.shape[1:]))]) if n != raw_size else x
    regions_h = regions_f.permute(1, 0, 2).reshape(2, n_tables * num_heads)  # "c a h -> a (c h)"
    return x_p, {"raw_size": raw_size, "coords": coords_pad.to(coords.dtype), "region_indices": [eta, phi],
                 "regions_h": regions_h}
