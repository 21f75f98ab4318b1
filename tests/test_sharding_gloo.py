"""Table sharding over ranks (hept_amd/sharding.py) on CPU: world_size 2, gloo backend.

The per-rank partials are produced by the ORACLE here (there is no GPU in this container and the
product has no CPU compute path); what is under test is the product's sharding logic: table slices,
the exchange step (all-reduce / reduce-scatter + all-gather / all-to-all of f32 or packed rows + all-gather),
point slices and the finishing call.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import hept_oracle as ho
from hept_amd.ops import unpack_part
from hept_amd.sharding import TableSharding, table_slice


def test_table_slice_partitions():
    for t in range(1, 12):
        for w in range(1, t + 1):
            got = [table_slice(t, r, w) for r in range(w)]
            assert got[0][0] == 0 and sum(c for _, c in got) == t
            assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(w - 1))
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        table_slice(2, 0, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _acc_from_oracle(inp, t0, tl):
    """(N, H, 32) [numer | denom | 0] summed over tables [t0, t0+tl), the layout the HIP path produces."""
    res = ho.forward_partials(
        inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"][t0:t0 + tl], inp["w_rpe_weight"],
        inp["alpha"][:, :, t0:t0 + tl].contiguous(), block_size=inp["block_size"], w_per_dist=inp["w_per_dist"], keep=False)
    numer, denom = res["numer"].sum(0), res["denom"].sum(0)  # (H,N,D), (H,N,1)
    h, n, d = numer.shape
    acc = torch.zeros(n, h, 32)
    acc[..., :d] = numer.permute(1, 0, 2)
    acc[..., d] = denom.squeeze(-1).permute(1, 0)
    return acc


def _pack_rows(acc):
    """f32 rows (N,H,32) -> the packed 64-B row format of the 16-bit HIP modes: (N,H,16) int32 =
    [24 bf16 numerators | f32 denominator | 0] (include/hept_hip.h, hept_part_precision)."""
    n, h, _ = acc.shape
    bits = acc[..., :24].to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    out = torch.zeros(n, h, 16, dtype=torch.int32)
    out[..., :12] = bits[..., 0::2] | (bits[..., 1::2] << 16)
    out[..., 12] = acc[..., 24].contiguous().view(torch.int32)
    return out


def _finish_cpu(inp):
    d = inp["out_weight"].shape[0]

    def fn(part, n0, cnt):
        part = unpack_part(part)          # packed rows -> f32 rows (no-op for f32)
        if part.dim() == 4:               # all_to_all: one slice per source rank, summed here
            part = part.sum(0)
        rows = part[n0:n0 + cnt]
        per_head = rows[..., :d] / rows[..., d:d + 1]
        return torch.nn.functional.linear(per_head.reshape(cnt, -1), inp["out_weight"], inp["out_bias"])

    return fn


def _worker(rank, world, port, mode, name, ret, groups=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        inp, _ = cases.load_case(name)
        n_tables = inp["alpha"].shape[2]
        sh = TableSharding(n_tables, dist.group.WORLD,
                           mode="all_to_all" if mode == "all_to_all_packed" or mode.startswith("pipelined") else mode)
        t0, tl = sh.local_tables()
        acc = _acc_from_oracle(inp, t0, tl)
        if mode.startswith("pipelined"):
            # the production exchange: head groups sent one at a time (TableSharding.pipelined); the producer and the
            # finishing call are CPU stand-ins for hept_partial_heads / hept_combine_groups
            if mode == "pipelined_packed":
                acc = _pack_rows(acc)
            n, h, row = acc.shape
            sh.mode, sh.head_groups = "all_to_all", groups
            d = inp["out_weight"].shape[0]
            seen = []

            def produce(g, h0, dst):
                seen.append((g, h0, tuple(dst.shape)))
                dst.zero_()
                dst[:n] = acc[:, h0:h0 + dst.shape[1]]

            def finish(recv, cnt):
                wide = unpack_part(recv).sum(1)                          # (G, per, hg, 32): summed over the ranks
                rows = wide.permute(1, 0, 2, 3).reshape(wide.shape[1], h, 32)[:cnt]
                per_head = rows[..., :d] / rows[..., d:d + 1]
                return torch.nn.functional.linear(per_head.reshape(cnt, -1), inp["out_weight"], inp["out_bias"])

            out = sh.pipelined(n, h, row, acc.dtype, acc.device, produce, finish)
            assert [s_[0] for s_ in seen] == list(range(sh.groups_for(h)))
            if rank == 0:
                ret["out"] = out.clone()
            gathered = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(gathered, out)
            if rank == 0:
                ret["same_on_all_ranks"] = all(torch.equal(gathered[0], g) for g in gathered)
            return
        packed = mode == "all_to_all_packed"
        if packed:
            acc, mode = _pack_rows(acc), "all_to_all"
            sh.mode = "all_to_all"
        if mode in ("reduce_scatter", "all_to_all"):
            # gloo has no reduce_scatter: emulate the collective pair so that the slicing / padding /
            # gather logic of TableSharding.finish still runs end to end
            orig_rs, orig_ag = dist.reduce_scatter_tensor, dist.all_gather_into_tensor

            def rs(out, inp_t, op=None, group=None):
                full = inp_t.clone()
                dist.all_reduce(full, group=group)
                per = out.shape[0]
                out.copy_(full[rank * per:(rank + 1) * per])

            def ag(out, inp_t, group=None):
                parts = [torch.empty_like(inp_t) for _ in range(world)]
                dist.all_gather(parts, inp_t, group=group)
                out.copy_(torch.cat(parts, 0))

            dist.reduce_scatter_tensor, dist.all_gather_into_tensor = rs, ag
        out = sh.finish(acc, _finish_cpu(inp))
        if mode in ("reduce_scatter", "all_to_all"):
            dist.reduce_scatter_tensor, dist.all_gather_into_tensor = orig_rs, orig_ag
        if rank == 0:
            ret["out"] = out.clone()
        gathered = [torch.empty_like(out) for _ in range(world)]
        dist.all_gather(gathered, out)
        if rank == 0:
            ret["same_on_all_ranks"] = all(torch.equal(gathered[0], g) for g in gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,name,world,groups", [
    ("pipelined", "g6_block100", 2, 2), ("pipelined_packed", "g6_block100", 2, 4), ("pipelined", "g4_pileup", 3, 2),
    ("pipelined_packed", "g4_pileup", 3, 8), ("pipelined", "g1_rand512", 2, 3)])
def test_pipelined_exchange_matches_single_process(mode, name, world, groups):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, name, ret, groups), nprocs=world, join=True)
    inp, _ = cases.load_case(name)
    assert ret["same_on_all_ranks"]
    if mode == "pipelined_packed":
        n_tables = inp["alpha"].shape[2]
        accs = [unpack_part(_pack_rows(_acc_from_oracle(inp, *table_slice(n_tables, r, world)))) for r in range(world)]
        want = _finish_cpu(inp)(torch.stack(accs), 0, accs[0].shape[0])
    else:
        want = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                          inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=inp["block_size"],
                          w_per_dist=inp["w_per_dist"], keep=False)["out"]
    torch.testing.assert_close(ret["out"], want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode,name,world", [
    ("all_reduce", "g6_block100", 2), ("reduce_scatter", "g6_block100", 2), ("all_reduce", "g1_rand512", 2),
    ("all_to_all", "g6_block100", 2), ("all_to_all_packed", "g6_block100", 2),
    # three ranks, one table each, and a point count (4096) that does not divide: the last slice is padded
    ("all_to_all", "g4_pileup", 3), ("all_to_all_packed", "g4_pileup", 3), ("reduce_scatter", "g4_pileup", 3)])
def test_table_sharding_matches_single_process(mode, name, world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, name, ret), nprocs=world, join=True)
    inp, _ = cases.load_case(name)
    assert ret["same_on_all_ranks"]
    if mode == "all_to_all_packed":
        # expected: every rank's sum rounded to the packed row format once, then summed and finished
        n_tables = inp["alpha"].shape[2]
        accs = [unpack_part(_pack_rows(_acc_from_oracle(inp, *table_slice(n_tables, r, world)))) for r in range(world)]
        want = _finish_cpu(inp)(torch.stack(accs), 0, accs[0].shape[0])
        torch.testing.assert_close(ret["out"], want, rtol=1e-5, atol=1e-6)
        return
    ref = ho.forward(inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"], inp["w_rpe_weight"],
                     inp["alpha"], inp["out_weight"], inp["out_bias"], block_size=inp["block_size"],
                     w_per_dist=inp["w_per_dist"], keep=False)["out"]
    # the sharded sum adds the same per-table terms in a different association: fp32 round-off only
    torch.testing.assert_close(ret["out"], ref, rtol=1e-5, atol=1e-6)
