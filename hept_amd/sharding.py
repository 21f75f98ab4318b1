"""Table sharding of the OR-LSH hash tables over the GPUs of one node.

The ``n_hashes`` tables are independent from the E2LSH projection up to the
per-table partials; the only coupling in the reference is
``o.sum(0) / logits.sum(0)`` (``example/hept.py:79``).  Rank r owns a contiguous
slice of tables, computes ``acc_r (N, H, 32) = sum_t [numer | denom]`` with the
HIP kernels, and the ranks then sum ``acc`` with ONE exchange step over
RCCL/xGMI:

* ``mode="reduce_scatter"`` (default on NCCL/RCCL): reduce-scatter over point
  slices -> every rank divides and applies ``out_linear`` on its N/G slice ->
  all-gather of the (N, D) output.  Moves (G-1)/G * (|acc| + |out|) per rank
  instead of 2 (G-1)/G |acc| for a ring all-reduce; xGMI is point-to-point, so
  per-link bytes are what matters.
* ``mode="all_reduce"``: all-reduce of ``acc``, every rank finishes all points
  (used on backends without reduce-scatter, e.g. gloo in the CPU tests).

This module is device-agnostic glue around ``torch.distributed``; the
numerator/denominator arithmetic stays in the HIP ``combine_out`` kernel, which
is passed in as ``finish_fn``.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["TableSharding", "table_slice"]


def table_slice(n_tables: int, rank: int, world: int) -> Tuple[int, int]:
    """(first table, count) owned by ``rank``: contiguous, sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_tables < world:
        raise ValueError(f"n_hashes={n_tables} cannot be sharded over {world} ranks (need >= 1 table per rank)")
    base, extra = divmod(n_tables, world)
    t0 = rank * base + min(rank, extra)
    return t0, base + (1 if rank < extra else 0)


class TableSharding:
    def __init__(self, n_tables: int, group=None, mode: Optional[str] = None, always_exchange: bool = False):
        self.group = group
        self.always_exchange = always_exchange  # run the collectives even on a 1-rank group (tests)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_tables = n_tables
        backend = dist.get_backend(group)
        self.mode = mode or ("reduce_scatter" if backend == "nccl" else "all_reduce")
        if self.mode not in ("reduce_scatter", "all_reduce"):
            raise ValueError(f"unknown mode {self.mode}")
        table_slice(n_tables, self.rank, self.world)  # validate

    def local_tables(self) -> Tuple[int, int]:
        return table_slice(self.n_tables, self.rank, self.world)

    def point_slice(self, n_points: int, rank: Optional[int] = None) -> Tuple[int, int]:
        """(first point, count) a rank finishes in reduce_scatter mode; slices are equal, the last may be short."""
        rank = self.rank if rank is None else rank
        per = (n_points + self.world - 1) // self.world
        n0 = min(rank * per, n_points)
        return n0, min(per, n_points - n0)

    def finish(self, acc: torch.Tensor, finish_fn: Callable[[torch.Tensor, int, int], torch.Tensor]) -> torch.Tensor:
        """Sum ``acc`` (N, H, 32) over the ranks; return the full (N, D) output on every rank.

        ``finish_fn(part, n0, count)`` turns summed partials into outputs: it reads rows
        ``part[n0 : n0 + count]`` and returns ``(count, D)`` (``count`` may be 0).
        """
        n = acc.shape[0]
        if self.world == 1 and not self.always_exchange:
            return finish_fn(acc, 0, n)
        if self.mode == "all_reduce":
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
            return finish_fn(acc, 0, n)
        per = (n + self.world - 1) // self.world
        padded = per * self.world
        if padded != n:  # reduce_scatter_tensor needs equal shards
            acc = torch.cat([acc, acc.new_zeros((padded - n,) + tuple(acc.shape[1:]))], dim=0)
        mine = torch.empty((per,) + tuple(acc.shape[1:]), device=acc.device, dtype=acc.dtype)
        dist.reduce_scatter_tensor(mine, acc, op=dist.ReduceOp.SUM, group=self.group)
        _, cnt = self.point_slice(n)
        out_slice = finish_fn(mine, 0, cnt)
        d = out_slice.shape[1]
        if cnt != per:
            out_slice = torch.cat([out_slice, out_slice.new_zeros((per - cnt, d))], dim=0)
        full = torch.empty((padded, d), device=acc.device, dtype=out_slice.dtype)
        dist.all_gather_into_tensor(full, out_slice.contiguous(), group=self.group)
        return full[:n]
