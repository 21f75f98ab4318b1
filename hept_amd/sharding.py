"""Table sharding of the OR-LSH hash tables over the GPUs of one node.

The ``n_hashes`` tables are independent from the E2LSH projection up to the
per-table partials; the only coupling in the reference is
``o.sum(0) / logits.sum(0)`` (``example/hept.py:79``).  Rank r owns a contiguous
slice of tables, computes ``acc_r (N, H, 32) = sum_t [numer | denom]`` with the
HIP kernels, and the ranks then sum ``acc`` with ONE exchange step over
RCCL/xGMI:

* ``mode="all_to_all"`` (default on NCCL/RCCL): every rank sends point slice g
  of its ``acc`` straight to rank g (one all-to-all: G-1 concurrent
  point-to-point transfers per rank, one per xGMI link, no ring), rank g sums
  the G slices it received inside the HIP ``combine_out`` kernel (they are read
  as G "tables"), divides, applies ``out_linear`` on its N/G points, and an
  all-gather distributes the (N, D) output.  ``acc`` travels in the row format
  the caller chose: with the 16-bit tile modes the packed 64-B rows (bf16
  numerators, f32 denominator) halve the bytes on the links.
* ``mode="reduce_scatter"``: the same split with RCCL's reduce-scatter doing the
  sum (f32 rows only).
* ``mode="all_reduce"``: all-reduce of ``acc``, every rank finishes all points
  (used on backends without reduce-scatter, e.g. gloo in the CPU tests).

The all-to-all is pipelined behind the block attention (``TableSharding.pipelined``): the heads are cut into
``head_groups`` groups; as soon as the block attention of group g has written its table-summed rows, a
communication stream sends them while the main stream computes group g + 1, and the HIP combine reads the
received groups in place (``hept_combine_groups``).  Only the last group's transfer, the combine of the N/G-point
slice and the all-gather of the (N, D) output stay on the critical path.

This module is device-agnostic glue around ``torch.distributed``; the
numerator/denominator arithmetic stays in the HIP ``combine_out`` kernel, which
is passed in as ``finish_fn``.
"""
from __future__ import annotations

import os
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
    def __init__(self, n_tables: int, group=None, mode: Optional[str] = None, always_exchange: bool = False,
                 head_groups: Optional[int] = None, out_view: bool = False):
        self.group = group
        # one-sided transport only, opt-in: the sharded forward returns a tensor over the exchange buffer instead of
        # copying the gathered output out of it (include/hept_hip.h, hept_comm_set_out_view).  The tensor is
        # overwritten by the SECOND next sharded forward on this object: not for callers that keep outputs around
        # (autograd, several HEPT layers sharing one TableSharding with a skip connection across two of them).
        self.out_view = bool(out_view)
        # head groups of the pipelined all-to-all; None = by transport (one-sided stores: 4 -- a group's push rides in
        # the next group's attention launch at no extra cost, and the last, exposed push is a quarter of the rows;
        # collectives: 2 -- every further group is another collective launch and cross-stream hand-over)
        env = os.environ.get("HEPT_HEAD_GROUPS")
        self.head_groups = int(head_groups) if head_groups is not None else (int(env) if env else None)
        self._bufs = {}
        self._comm = None
        # hept_comm* of the native exchange (hept_forward_sharded: RCCL called from the C library, no Python between
        # the kernels and the collectives); None = not tried yet, 0 = unavailable (torch.distributed path)
        self._native = None
        self._xbuf = None
        self._p2p_failed = False
        # transport of the all-to-all exchange: "p2p" (one-sided xGMI stores; falls back to "rccl" when the peers'
        # buffers cannot be mapped), "rccl" (RCCL called from the C library), "torch" (torch.distributed collectives)
        self.exchange = os.environ.get("HEPT_EXCHANGE", "p2p")
        self.always_exchange = always_exchange  # run the collectives even on a 1-rank group (tests)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_tables = n_tables
        backend = dist.get_backend(group)
        self.mode = mode or ("all_to_all" if backend == "nccl" else "all_reduce")
        if self.mode not in ("all_to_all", "reduce_scatter", "all_reduce"):
            raise ValueError(f"unknown mode {self.mode}")
        if self.mode == "all_to_all" and self.world > 16:
            raise ValueError("all_to_all mode serves the ranks of one node (at most 16)")
        table_slice(n_tables, self.rank, self.world)  # validate

    def describe(self) -> str:
        if self.mode == "all_to_all":
            via = "torch.distributed"
            if self._native:
                via = "RCCL from the C library" if (self._p2p_failed or self.exchange == "rccl") else "one-sided xGMI stores"
            return f"all_to_all pipelined in {self.head_groups or 'auto'} head group(s) + all_gather ({via})"
        return self.mode

    def _agree(self, ok: bool, device: torch.device) -> bool:
        """True iff ``ok`` on every rank (a collective)."""
        on = device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if ok else 0], device=on, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t) == 1

    def native_comm(self, device: torch.device) -> int:
        """``hept_comm*`` for the one-call exchange, or 0 when it is not available (``HEPT_EXCHANGE=torch``, CPU
        tensors, or communicator creation failed on some rank).  On an RCCL process group the communicator carries
        its own RCCL communicator; on any other backend (gloo: several ranks sharing one GPU in the tests) it has the
        one-sided transport only."""
        if self._native is not None:
            return self._native
        self._native = 0
        if self.mode != "all_to_all" or self.exchange == "torch" or device.type != "cuda" or self.world > 16:
            return 0
        import ctypes

        from . import _lib

        lib = _lib.load()
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            if dist.get_backend(self.group) == "nccl":
                ident = (ctypes.c_char * 128)()
                box = [None]
                if self.rank == 0 and lib.hept_comm_unique_id(ident) == 0:
                    box[0] = bytes(ident.raw)
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast_object_list(box, src=src, group=self.group, device=device)
                rc = 1
                if box[0] is not None:
                    rc = lib.hept_comm_create(box[0], self.rank, self.world, ctypes.byref(handle))
            else:
                rc = lib.hept_comm_create_local(self.rank, self.world, ctypes.byref(handle))
        if self._agree(rc == 0, device):
            self._native = handle.value
        else:
            if rc == 0:
                lib.hept_comm_destroy(handle)
            import warnings

            warnings.warn("hept_amd: native communicator unavailable ("
                          + (lib.hept_comm_last_error() or b"").decode(errors="replace")
                          + "); table sharding falls back to torch.distributed collectives")
        return self._native

    def one_sided(self, nbytes: int, device: torch.device) -> bool:
        """Whether the one-sided transport is set up for exchange buffers of ``nbytes`` (sets it up on first use:
        allocate, swap the IPC handles through the process group, map the peers -- a collective)."""
        comm = self._native
        if not comm or self.exchange == "rccl" or self._p2p_failed:
            return False
        from . import _lib

        lib = _lib.load()
        if lib.hept_comm_p2p_ready(comm, nbytes):
            return True
        import ctypes

        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(device):
            rc = lib.hept_comm_p2p_alloc(comm, nbytes, handle)
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (rc, handle.raw), group=self.group)
            ok = all(r == 0 for r, _ in gathered)
            if ok:
                ok = lib.hept_comm_p2p_open(comm, b"".join(hb for _, hb in gathered)) == 0
        if self._agree(ok, device):
            lib.hept_comm_set_out_view(comm, 1 if self.out_view else 0)
            # every rank's buffer (and its zeroed flags) exists before anybody stores into it
            dist.barrier(group=self.group)
            return True
        self._p2p_failed = True
        import warnings

        warnings.warn("hept_amd: one-sided exchange unavailable ("
                      + (lib.hept_comm_last_error() or b"").decode(errors="replace") + "); using "
                      + ("RCCL collectives" if lib.hept_comm_has_rccl(comm) else "torch.distributed collectives"))
        if not lib.hept_comm_has_rccl(comm):
            lib.hept_comm_destroy(comm)
            self._native = 0
        return False

    def downgrade(self) -> bool:
        """Step down one rung of the transport ladder (one-sided stores -> RCCL from the C library -> torch.distributed
        all-to-all -> reduce-scatter); False when there is nothing left.  Every rank must take the same step."""
        from . import _lib

        lib = _lib.load()
        if self._native and self.exchange != "rccl" and lib.hept_comm_has_rccl(self._native):
            self.exchange = "rccl"   # (also after a one-sided timeout: the communicator's RCCL half is intact)
            return True
        if self._native or self._native is None:
            if self._native:
                lib.hept_comm_destroy(self._native)
            self._native, self.exchange = 0, "torch"
            return True
        if self.mode == "all_to_all" and dist.get_backend(self.group) == "nccl":
            self.mode = "reduce_scatter"
            return True
        return False

    def tune(self, step: Callable[[], object], device: torch.device, head_groups=(1, 2, 4, 8), steps: int = 10):
        """Pick the fastest (transport, head groups) for this machine: time ``step`` (one sharded forward) under every
        candidate the communicator supports and keep the best.  Collective: all ranks time the same candidates, the
        slowest rank's time counts, a candidate that fails or times out on any rank is dropped.  Returns the table
        {(transport, groups): seconds per step}."""
        import time

        from . import _lib

        lib = _lib.load()
        if not self._native:
            return {}
        transports = []
        if self.exchange == "p2p" and not self._p2p_failed:
            transports.append("p2p")
        if lib.hept_comm_has_rccl(self._native):
            transports.append("rccl")
        on = device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        table = {}
        for tr in transports:
            for g in head_groups:
                self.exchange, self.head_groups = tr, g
                bad, dt = 0, 0.0

                def run(count):
                    """`count` steps; a failure on this rank is remembered, never raised past here: the sequence of
                    collectives below (barrier, all_reduce) must be the same on every rank whatever happened"""
                    nonlocal bad
                    try:
                        for _ in range(count):
                            if not bad:
                                step()
                        torch.cuda.synchronize(device)
                    except Exception:  # noqa: BLE001
                        bad = 1

                run(3)
                dist.barrier(group=self.group)
                t0 = time.perf_counter()
                run(steps)
                dt = (time.perf_counter() - t0) / steps
                try:
                    self.check()
                except Exception:  # noqa: BLE001
                    bad = 1
                t = torch.tensor([dt, float(bad)], device=on, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                if t[1].item() == 0:
                    table[(tr, g)] = t[0].item()
                elif tr == "p2p":
                    # on every rank (the flag was reduced): no further one-sided candidates; the reset is collective
                    # (epochs and flags restart together), and nobody stores into a buffer that is being cleared
                    self._p2p_failed = True
                    lib.hept_comm_reset_status(self._native)
                    dist.barrier(group=self.group)
                    break
        if table:
            self.exchange, self.head_groups = min(table, key=table.get)
        return table

    def check(self) -> None:
        """Raise if a one-sided wait has timed out (synchronises the device; call it outside timed regions)."""
        if self._native:
            import ctypes

            from . import _lib

            st = ctypes.c_int(0)
            _lib.check(_lib.load().hept_comm_status(self._native, ctypes.byref(st)), "hept_comm_status")
            if st.value:
                flags = (ctypes.c_uint32 * 1024)()
                epoch = ctypes.c_uint32(0)
                _lib.load().hept_comm_p2p_flags(self._native, flags, ctypes.byref(epoch))
                rows = [list(flags[g * 16:g * 16 + self.world]) for g in range(self.groups_for(8))]
                # the transport is not used again by this object, and the recorded timeout is forgotten so that the
                # transports further down the ladder are not blamed for it
                self._p2p_failed = True
                _lib.load().hept_comm_reset_status(self._native)
                raise RuntimeError(f"hept_amd: one-sided exchange timed out waiting for a peer (status {st.value}; "
                                   f"rank {self.rank} epoch {epoch.value}, row flags {rows}, output flags "
                                   f"{list(flags[512:512 + self.world])})")

    def exchange_buffer(self, nbytes: int, device: torch.device) -> torch.Tensor:
        if self._xbuf is None or self._xbuf.numel() < nbytes or self._xbuf.device != device:
            self._xbuf = torch.empty(nbytes, device=device, dtype=torch.uint8)
        return self._xbuf

    def close(self) -> None:
        if getattr(self, "_live_views", 0) > 0 and self._native:
            # (every live view holds a reference to this object, so __del__ cannot get here with views alive)
            raise RuntimeError(f"TableSharding.close(): {self._live_views} view(s) of the gathered output (out_view=True) "
                               "are still alive; drop them first -- the exchange buffer they point into would be unmapped")
        if self._native:
            from . import _lib

            _lib.load().hept_comm_destroy(self._native)
        self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def groups_for(self, n_heads: int) -> int:
        """Head groups actually used for ``n_heads`` heads (equal groups only).  One-sided transport with ONE local
        table: the block attention stores its rows straight into the owners' buffers while it runs, so one launch
        (one group) already overlaps the transfer with the computation."""
        want = self.head_groups
        if want is None:
            p2p = bool(self._native and self.exchange == "p2p" and not self._p2p_failed)
            # (every rank must arrive at the same count: the receive buffers are laid out by head group -- so the
            #  choice rests on the table count of the whole group, not on this rank's share)
            one_each = self.n_tables == self.world
            want = (1 if one_each else 4) if p2p else 2
        g = max(1, min(want, n_heads))
        while n_heads % g != 0:
            g -= 1
        return g

    def _all_gather_rows(self, full: torch.Tensor, mine: torch.Tensor) -> None:
        if dist.get_backend(self.group) == "gloo":  # gloo has no all_gather_into_tensor
            per = mine.shape[0]
            parts = [full[r * per:(r + 1) * per] for r in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
        else:
            dist.all_gather_into_tensor(full, mine, group=self.group)

    def pipelined(self, n: int, n_heads: int, row: int, dtype: torch.dtype, device: torch.device,
                  produce: Callable[[int, int, torch.Tensor], None],
                  finish_fn: Callable[[torch.Tensor, int], torch.Tensor]) -> torch.Tensor:
        """The all-to-all exchange with its transfers hidden behind the producer.

        ``produce(g, h0, dst)`` fills ``dst`` (world * per, hg, row) with this rank's table-summed rows of heads
        [h0, h0 + hg) on the current stream; group g is sent (rank r gets points [r * per, (r + 1) * per)) while
        ``produce`` runs for group g + 1.  ``finish_fn(recv, count)`` gets ``recv`` (G, world, per, hg, row) -- for
        every head group the slices of all ranks -- sums over the ranks and returns ``(count, D)`` for the first
        ``count`` points of this rank's slice.  Returns the full (N, D) output on every rank.
        """
        ng = self.groups_for(n_heads)
        hg = n_heads // ng
        per = (n + self.world - 1) // self.world
        key = (n, n_heads, row, dtype, device, ng)
        bufs = self._bufs.get(key)
        if bufs is None:
            self._bufs.clear()  # one shape at a time: the buffers are as large as the partial rows
            send = torch.empty((ng, self.world * per, hg, row), device=device, dtype=dtype)
            recv = torch.empty((ng, self.world, per, hg, row), device=device, dtype=dtype)
            events = [torch.cuda.Event() for _ in range(ng)] if device.type == "cuda" else None
            bufs = self._bufs[key] = (send, recv, events)
        send, recv, events = bufs
        on_gpu = device.type == "cuda"
        if on_gpu and self._comm is None:
            self._comm = torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream(device) if on_gpu else None
        for g in range(ng):
            produce(g, g * hg, send[g])
            if on_gpu:
                events[g].record(main)
                self._comm.wait_event(events[g])
                with torch.cuda.stream(self._comm):
                    dist.all_to_all_single(recv[g].view(self.world * per, hg, row), send[g], group=self.group)
            else:
                dist.all_to_all_single(recv[g].view(self.world * per, hg, row), send[g], group=self.group)
        if on_gpu:
            main.wait_stream(self._comm)
        _, cnt = self.point_slice(n)
        out_slice = finish_fn(recv, cnt)
        d = out_slice.shape[1]
        if cnt != per:
            out_slice = torch.cat([out_slice, out_slice.new_zeros((per - cnt, d))], dim=0)
        full = torch.empty((self.world * per, d), device=device, dtype=out_slice.dtype)
        self._all_gather_rows(full, out_slice.contiguous())
        return full[:n]

    def local_tables(self) -> Tuple[int, int]:
        return table_slice(self.n_tables, self.rank, self.world)

    def point_slice(self, n_points: int, rank: Optional[int] = None) -> Tuple[int, int]:
        """(first point, count) a rank finishes in reduce_scatter mode; slices are equal, the last may be short."""
        rank = self.rank if rank is None else rank
        per = (n_points + self.world - 1) // self.world
        n0 = min(rank * per, n_points)
        return n0, min(per, n_points - n0)

    @property
    def packed_ok(self) -> bool:
        """Whether ``finish`` accepts packed (int32) rows: only the all-to-all moves them without arithmetic."""
        return self.mode == "all_to_all"

    def finish(self, acc: torch.Tensor, finish_fn: Callable[[torch.Tensor, int, int], torch.Tensor]) -> torch.Tensor:
        """Sum ``acc`` (N, H, row) over the ranks; return the full (N, D) output on every rank.

        ``finish_fn(part, n0, count)`` turns partial rows into outputs: ``part`` is (N', H, row) or a stack
        (G, N', H, row) whose leading dimension it has to sum; it reads rows ``[n0, n0 + count)`` and returns
        ``(count, D)`` (``count`` may be 0).
        """
        n = acc.shape[0]
        if self.world == 1 and not self.always_exchange:
            return finish_fn(acc, 0, n)
        if self.mode != "all_to_all" and acc.dtype != torch.float32:
            raise TypeError(f"mode {self.mode} sums inside the collective and needs f32 rows")
        if self.mode == "all_reduce":
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
            return finish_fn(acc, 0, n)
        per = (n + self.world - 1) // self.world
        padded = per * self.world
        if padded != n:  # the collectives need equal shards
            acc = torch.cat([acc, acc.new_zeros((padded - n,) + tuple(acc.shape[1:]))], dim=0)
        _, cnt = self.point_slice(n)
        if self.mode == "all_to_all":
            recv = torch.empty((self.world, per) + tuple(acc.shape[1:]), device=acc.device, dtype=acc.dtype)
            dist.all_to_all_single(recv, acc, group=self.group)  # recv[g] = rank g's rows of my point slice
            out_slice = finish_fn(recv, 0, cnt)
        else:
            mine = torch.empty((per,) + tuple(acc.shape[1:]), device=acc.device, dtype=acc.dtype)
            dist.reduce_scatter_tensor(mine, acc, op=dist.ReduceOp.SUM, group=self.group)
            out_slice = finish_fn(mine, 0, cnt)
        d = out_slice.shape[1]
        if cnt != per:
            out_slice = torch.cat([out_slice, out_slice.new_zeros((per - cnt, d))], dim=0)
        full = torch.empty((padded, d), device=acc.device, dtype=out_slice.dtype)
        self._all_gather_rows(full, out_slice.contiguous())
        return full[:n]
