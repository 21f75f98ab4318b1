"""``HEPTAttention``: drop-in ``nn.Module`` for the reference operator, backed by the gfx950 HIP kernels.

Mirrors the reference interface (``example/hept.py:31-81``): same constructor
(``hash_dim`` positional, model-config kwargs ``h_dim, num_heads, block_size,
n_hashes, num_w_per_dist``; extra keys ignored), same ``forward(query, key,
value, **kwargs)`` with ``kwargs["w_rpe"]`` (an ``nn.Linear``; only ``.weight``
is read), ``kwargs["coords"]`` (N, C), ``kwargs["combined_shifts"]`` (T, H, N)
int64, optional ignored ``pe``; same attribute and state-dict names
(``out_linear.weight``, ``out_linear.bias``, ``e2lsh.alpha``), so a reference
checkpoint loads with ``strict=True``.

The same class also stands in for the reference's ``src`` variant (``src/models/attention/hept.py:59-117``,
SURVEY.md §8 f-3): construct with ``variant="src"`` (adds the unused ``e2lsh.beta`` parameter that variant's
checkpoints carry) and call ``forward`` with that variant's kwargs — ``raw_size``, ``regions_h``,
``region_indices`` as built by ``hept_amd.prep.prepare_input_src`` — instead of ``combined_shifts``.

Two keyword-only extensions: ``precision`` ("fp32" reference numerics: f32 rows, tile products as split-bf16
MFMAs accurate to f32 round-off / "fp32_mfma" the same on the native f32 MFMA, slower, kept as ground truth /
"fp32_diff" = "fp32_mfma" with the coordinate part of the logit as explicit differences: the mode for a trained
``w_rpe`` on un-normalised coordinates, where the reference's own fp32 logits are rounding noise /
"bf16" MFMA tiles / "mixed16" = fp16 q̂,k̂ tiles with bf16 weights and values) and ``process_group`` (shard the ``n_hashes`` tables over
the ranks of a ``torch.distributed`` group).

Inference (``torch.no_grad()``) runs the whole operator in one C call.  When gradients are
required the module routes through ``hept_amd.autograd`` (HIP forward + HIP backward of the
block attention, f32 tiles).  There is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .sharding import TableSharding

__all__ = ["E2LSH", "HEPTAttention"]


class E2LSH(nn.Module):
    """Holds the frozen Gaussian projection ``alpha`` (H, E, T); reference ``example/hept_utils.py:38-47``.

    ``forward`` keeps the reference meaning (vecs (H,N,E) -> hashes (T,H,N)) for callers that
    use it on its own; the fused operator reads ``alpha`` directly.
    """

    def __init__(self, n_hashes, n_heads, dim, r=1, with_beta: bool = False):
        super().__init__()
        self.alpha = nn.Parameter(torch.normal(0, 1, (n_heads, dim, n_hashes)))
        self.alpha.requires_grad = False
        if with_beta:  # src/models/model_utils/hash_utils.py:322: drawn, stored in checkpoints, never read
            self.beta = nn.Parameter(r * torch.rand(1, n_hashes))
            self.beta.requires_grad = False

    def forward(self, vecs):
        return torch.bmm(vecs, self.alpha).permute(2, 0, 1)


class HEPTAttention(nn.Module):
    def __init__(self, hash_dim, *, precision: str = "fp32", process_group=None, variant: str = "example", **kwargs):
        super().__init__()
        if variant not in ("example", "src"):
            raise ValueError(f"variant must be 'example' or 'src', got {variant!r}")
        self.variant = variant
        self.dim_per_head = kwargs["h_dim"]
        self.num_heads = kwargs["num_heads"]
        self.out_linear = nn.Linear(self.num_heads * self.dim_per_head, self.dim_per_head)

        self.block_size = kwargs["block_size"]
        self.n_hashes = kwargs["n_hashes"]
        self.num_w_per_dist = kwargs["num_w_per_dist"]
        self.e2lsh = E2LSH(n_hashes=self.n_hashes, n_heads=self.num_heads, dim=hash_dim, with_beta=variant == "src")

        c_dim = hash_dim - self.dim_per_head
        if not (1 <= self.num_heads <= 16 and 1 <= self.dim_per_head <= 27 and c_dim >= 2 and hash_dim <= 30
                and 1 <= self.block_size <= 256 and self.n_hashes >= 1):
            raise ValueError(
                f"hept_amd.HEPTAttention supports 1 <= num_heads <= 16, 1 <= h_dim <= 27, coords_dim >= 2 with "
                f"h_dim + coords_dim <= 30 (32-column rows), 1 <= block_size <= 256 and any n_hashes >= 1; got "
                f"num_heads={self.num_heads}, h_dim={self.dim_per_head}, coords_dim={c_dim}, "
                f"block_size={self.block_size}, n_hashes={self.n_hashes}")
        self.precision = precision
        ops.precision_code(precision)  # validate early
        self.sharding: Optional[TableSharding] = (
            TableSharding(self.n_hashes, process_group) if process_group is not None else None
        )
        self._workspace: Optional[torch.Tensor] = None
        self._ws_stream_ptr = None        # stream of the last forward that used the workspace (see _scratch)
        self._busy = False                # a forward of this instance is being issued (two host threads at once: refused)
        self._warned_eval_grad = False

    def _scratch(self, nbytes: int, device) -> torch.Tensor:
        ws = self._workspace
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
            self._workspace = ws
        # One workspace per module: a forward issued on ANOTHER stream than the one before it would overwrite rows the
        # earlier forward may still be reading.  Enforced, not only documented (round 6): the new stream waits for the
        # stream of the call before it -- one event wait when the stream changes, nothing when it does not.
        ptr = ops.current_stream_ptr(device)     # the raw hipStream_t (one C call; no Stream object on the common path)
        last = self._ws_stream_ptr
        if last is not None and last != ptr:
            torch.cuda.current_stream(device).wait_stream(torch.cuda.ExternalStream(last, device=device))
        self._ws_stream_ptr = ptr                # (also handed to the C call: one stream lookup per forward, not two)
        return ws

    def reserve(self, n_points: int, n_coords: int, device) -> None:
        """Allocate the inference workspace for clouds of up to ``n_points`` (padded) points now, so the first forward
        does not pay for a device allocation (tens of milliseconds for the 60k-point workspace); optional -- forward()
        grows the workspace on demand."""
        tl = self.n_hashes if self.sharding is None else self.sharding.local_tables()[1]
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._scratch(ops.workspace_bytes(int(n_points), self.num_heads, self.dim_per_head, int(n_coords), tl,
                                          self.block_size, self.precision), device)

    def forward(self, query, key, value, **kwargs):
        if self._busy:   # (a flag, not a lock: the point is to refuse a second host thread, not to queue it)
            raise RuntimeError("hept_amd.HEPTAttention: this instance is being called from two threads at once; it owns "
                               "one workspace -- use one instance per thread")
        self._busy = True
        try:
            return self._forward_impl(query, key, value, kwargs)
        finally:
            self._busy = False

    def _forward_impl(self, query, key, value, kwargs):
        if not query.is_cuda:
            raise RuntimeError("hept_amd.HEPTAttention needs GPU tensors: there is no CPU fallback")
        if torch.is_grad_enabled() and any(
            t.requires_grad for t in (query, key, value, kwargs["w_rpe"].weight, self.out_linear.weight)
        ):
            # differentiable path (f32 tiles).  The usual ``model.eval(); model(x)`` without ``torch.no_grad()`` also
            # lands here because the parameters require grad: with fp32 tiles on one GPU it simply takes the autograd
            # path (same values); with 16-bit tiles or table sharding an eval-mode call whose inputs carry no
            # gradient is served by the inference path instead, as the reference would serve it
            light = self.precision in ("fp32", "fp32_mfma", "fp32_diff") and self.sharding is None
            if light or self.training or any(t.requires_grad for t in (query, key, value)):
                return self._forward_train(query, key, value, **kwargs)
            if not self._warned_eval_grad:
                import warnings

                warnings.warn("hept_amd.HEPTAttention: eval-mode call with grad enabled and no input gradient under "
                              f"precision={self.precision!r}/table sharding runs the inference path (no parameter "
                              "gradients); call under torch.no_grad() or .train() to be explicit", stacklevel=2)
                self._warned_eval_grad = True
        if self.sharding is not None and self.sharding.out_view and torch.is_grad_enabled():
            # a view of the exchange buffer is overwritten by the second next call: autograd must never save it
            raise RuntimeError("TableSharding(out_view=True) hands out views of the exchange buffer that later calls "
                               "overwrite: call the module under torch.no_grad()")
        coords = kwargs["coords"]
        src = "combined_shifts" not in kwargs  # the src variant's kwargs: raw_size, regions_h, region_indices
        w_rpe_weight = kwargs["w_rpe"].weight
        n = query.shape[0]
        if n % self.block_size != 0:
            raise ValueError(f"number of points {n} is not a multiple of block_size {self.block_size}")
        h, d, c = self.num_heads, self.dim_per_head, coords.shape[1]
        common = dict(block_size=self.block_size, w_per_dist=self.num_w_per_dist, precision=self.precision)
        with torch.no_grad():
            # (w_rpe.weight goes to the C call as it is: sqrt_w (H, C), reference example/hept.py:22-25, is computed in
            #  the row builder's prologue on every forward -- nothing derived from a parameter is cached here)
            f32 = torch.float32
            q2 = query if (query.dtype is f32 and query.dim() == 2) else query.reshape(n, h * d).float()
            k2 = key if (key.dtype is f32 and key.dim() == 2) else key.reshape(n, h * d).float()
            v2 = value if (value.dtype is f32 and value.dim() == 2) else value.reshape(n, h * d).float()
            if self.sharding is None and torch.compiler.is_compiling():
                # one opaque graph node instead of a ctypes call Dynamo cannot trace (hept_amd/library.py)
                from .library import forward_op, forward_src_op

                if src:
                    eta, phi = kwargs["region_indices"]
                    out = forward_src_op(q2, k2, v2, coords.float(), eta, phi, kwargs["regions_h"],
                                         int(kwargs["raw_size"]), w_rpe_weight, self.e2lsh.alpha,
                                         self.out_linear.weight, self.out_linear.bias, self.block_size,
                                         self.num_w_per_dist, self.precision)
                else:
                    out = forward_op(q2, k2, v2, coords.float(), kwargs["combined_shifts"], w_rpe_weight,
                                     self.e2lsh.alpha, self.out_linear.weight, self.out_linear.bias, self.block_size,
                                     self.num_w_per_dist, self.precision)
            elif self.sharding is None:
                ws = self._scratch(ops.workspace_bytes(n, h, d, c, self.n_hashes, self.block_size, self.precision),
                                   query.device)
                if src:
                    out = ops.forward_src(q2, k2, v2, coords.float(), kwargs["region_indices"], kwargs["regions_h"],
                                          kwargs["raw_size"], w_rpe_weight, self.e2lsh.alpha, self.out_linear.weight,
                                          self.out_linear.bias, workspace=ws, **common)
                else:
                    out = ops.forward(q2, k2, v2, coords.float(), kwargs["combined_shifts"], w_rpe_weight,
                                      self.e2lsh.alpha, self.out_linear.weight, self.out_linear.bias, workspace=ws,
                                      stream=self._ws_stream_ptr, **common)
            else:
                out = self._forward_sharded(q2, k2, v2, coords.float(), w_rpe_weight, src, kwargs, common)
        return out if query.dtype is torch.float32 else out.to(query.dtype)

    def _forward_sharded(self, q2, k2, v2, coords, w_rpe_weight, src, kwargs, common):
        """Tables [t0, t0 + tl) of this rank, then the exchange (reference coupling: example/hept.py:79)."""
        sh = self.sharding
        n, h, d = q2.shape[0], self.num_heads, self.dim_per_head
        c = coords.shape[1]
        t0, tl = sh.local_tables()
        packed = sh.packed_ok and ops.packed_partials(self.precision, d)
        ws = self._scratch(ops.workspace_bytes(n, h, d, c, tl, self.block_size, self.precision), q2.device)
        geo = (kwargs["region_indices"], kwargs["regions_h"], kwargs["raw_size"]) if src else None
        if sh.mode == "all_to_all" and (sh.world > 1 or sh.always_exchange):
            comm = sh.native_comm(q2.device)
            one_sided = bool(comm) and sh.one_sided(ops.p2p_bytes(n, h, d, sh.world, self.precision), q2.device)
            comm = sh.native_comm(q2.device)   # (a communicator without RCCL is dropped when the mapping failed)
            if comm:
                # one C call: kernels on this stream, finished head groups travel behind the block attention
                # (one-sided xGMI stores, or RCCL on the communicator's side stream), combine of this rank's points,
                # gather of the output
                xbuf = None if one_sided else sh.exchange_buffer(
                    ops.exchange_bytes(n, h, d, sh.world, self.precision), q2.device)
                return ops.forward_sharded(q2, k2, v2, coords, None if src else kwargs["combined_shifts"], w_rpe_weight,
                                           self.e2lsh.alpha, self.out_linear.weight, self.out_linear.bias, comm=comm,
                                           world=sh.world, t0=t0, tl=tl, head_groups=sh.groups_for(h), workspace=ws,
                                           xbuf=xbuf, one_sided=one_sided, geo=geo, out_view=sh.out_view, view_owner=sh,
                                           **common)
            # the same pipeline driven from Python over torch.distributed (gloo in the tests; fallback)
            dims = ops.partial_begin(q2, k2, v2, coords, None if src else kwargs["combined_shifts"], w_rpe_weight,
                                     self.e2lsh.alpha, t0=t0, tl=tl, workspace=ws, geo=geo, **common)
            return sh.pipelined(
                n, h, 16 if packed else 32, torch.int32 if packed else torch.float32, q2.device,
                lambda g, h0, dst: ops.partial_heads(ws, dims, tl, self.block_size, self.precision, h0, dst),
                lambda recv, cnt: ops.combine_groups(recv, d, self.out_linear.weight, self.out_linear.bias, 0, cnt))
        if src:
            acc = ops.forward_partial_src(q2, k2, v2, coords, kwargs["region_indices"], kwargs["regions_h"],
                                          kwargs["raw_size"], w_rpe_weight, self.e2lsh.alpha, t0=t0, tl=tl,
                                          workspace=ws, packed=packed, **common)
        else:
            acc = ops.forward_partial(q2, k2, v2, coords, kwargs["combined_shifts"], w_rpe_weight, self.e2lsh.alpha,
                                      t0=t0, tl=tl, workspace=ws, packed=packed, **common)
        return sh.finish(acc, lambda part, n0, cnt: ops.combine_out(part, d, self.out_linear.weight,
                                                                     self.out_linear.bias, n0, cnt))

    # Tiles of the TRAINING path: "fp32" (default: the reference's arithmetic, gradients pinned on its autograd) or
    # "bf16" (opt-in: the bf16 forward's rows and one bf16 MFMA per product in the backward as well -- about half the
    # step time, gradients to the accuracy of a bf16 forward).  Set on the instance or the class.
    train_tiles = "fp32"

    def _train_tiles(self) -> str:
        if self.train_tiles not in ("fp32", "bf16"):
            raise ValueError("train_tiles must be 'fp32' or 'bf16'")
        return "fp32" if self.precision == "fp32_mfma" else self.train_tiles

    def _train_fused_ok(self, x, kwargs) -> bool:
        """Whether the training-mode ``Attn`` block may hand its LayerNorm and projections to the fused row builder"""
        return (x.is_cuda and self.dim_per_head == 24 and self.num_heads == 8 and self.sharding is None
                and "combined_shifts" in kwargs and kwargs["coords"].shape[1] in (2, 4, 6)
                and x.dim() == 2 and x.shape[0] % self.block_size == 0)

    def _forward_train_fused(self, x, norm1, w_q, w_k, w_v, **kwargs):
        """Training mode of the ``Attn`` block (``hept_amd.Attn``): LayerNorm + the three projections + the operator as
        one autograd node whose forward never materialises q, k, v (``autograd.HeptPartialSumsFused``); returns the
        operator's output (N, D), reference ``example/transformer.py:155-159``."""
        from .autograd import HeptCombine, HeptPartialSumsFused, RpeScale

        h, d = self.num_heads, self.dim_per_head
        sqrt_w = RpeScale.apply(kwargs["w_rpe"].weight.float(), h, d, self.num_w_per_dist)
        acc = HeptPartialSumsFused.apply(x.float(), norm1.weight.float(), norm1.bias.float(), norm1.eps, w_q.weight.float(),
                                         w_k.weight.float(), w_v.weight.float(), kwargs["coords"].float(), sqrt_w,
                                         self.e2lsh.alpha.detach(), kwargs["combined_shifts"], self.block_size,
                                         self.precision == "fp32_mfma", self._train_tiles())
        return HeptCombine.apply(acc, self.out_linear.weight, self.out_linear.bias).to(x.dtype)

    def _forward_train(self, query, key, value, **kwargs):
        """Differentiable path: HIP forward/backward of the block attention inside autograd (f32 tiles)."""
        from .autograd import HeptPartialSums, RpeScale, ReplicatedGrad, sum_over_ranks

        # training runs f32 tiles -- also for a module whose inference precision is 16-bit -- so that the gradients are
        # those of the reference's fp32 arithmetic; `train_tiles = "bf16"` opts into the 16-bit kernels in both directions
        f32_mfma = self.precision == "fp32_mfma"
        tiles = self._train_tiles()
        n = query.shape[0]
        if n % self.block_size != 0:
            raise ValueError(f"number of points {n} is not a multiple of block_size {self.block_size}")
        h, d = self.num_heads, self.dim_per_head
        coords = kwargs["coords"].float()
        sqrt_w = RpeScale.apply(kwargs["w_rpe"].weight.float(), h, d, self.num_w_per_dist)
        geo = None
        if "combined_shifts" not in kwargs:
            geo = ops.geo_args(kwargs["region_indices"], kwargs["regions_h"], self.n_hashes, h, n) + (
                int(kwargs["raw_size"]),)
        q2, k2, v2 = (x.reshape(n, h * d).float() for x in (query, key, value))
        alpha, codes = self.e2lsh.alpha.detach(), kwargs.get("combined_shifts")
        sh = self.sharding
        if sh is not None and sh.world > 1:
            # table sharding under autograd: inputs and loss are replicated, so every rank differentiates its own
            # tables' partial sums; the forward sums them over the ranks, the backward sums the ranks' contributions
            # to the gradients of the replicated inputs (reference coupling example/hept.py:79)
            t0, tl = sh.local_tables()
            alpha = alpha[:, :, t0:t0 + tl].contiguous()
            if codes is not None:
                codes = codes[t0:t0 + tl].contiguous()
            if geo is not None:
                eta, phi, cfac, raw = geo
                geo = (eta.view(self.n_hashes, h, n)[t0:t0 + tl].reshape(tl * h, n).contiguous(),
                       phi.view(self.n_hashes, h, n)[t0:t0 + tl].reshape(tl * h, n).contiguous(),
                       cfac.view(self.n_hashes, h)[t0:t0 + tl].reshape(tl * h).contiguous(), raw)
            q2, k2, v2, coords, sqrt_w = (ReplicatedGrad.apply(x, sh.group) for x in (q2, k2, v2, coords, sqrt_w))
        acc = HeptPartialSums.apply(q2, k2, v2, coords, sqrt_w, alpha, codes, self.block_size, geo, f32_mfma, tiles)
        if sh is not None and sh.world > 1:
            acc = sum_over_ranks(acc, sh.group)
        from .autograd import HeptCombine

        # example/hept.py:79-80, HIP in both directions for every supported head count / head dimension
        out = HeptCombine.apply(acc, self.out_linear.weight, self.out_linear.bias)
        return out.to(query.dtype)
