#!/usr/bin/env python3
"""bench.py — HEPT attention-forward throughput on MI355X (BASELINE.json metric).

One "step" = one ``HEPTAttention.forward`` (the whole hot path: E2LSH hash -> sort -> block
attention -> combine + out_linear) over one synthetic tracking-60k cloud whose inputs are already
resident in HBM.  N = 1: BASELINE config 3 (N_raw 60000 -> 60032 padded, block 128, n_hashes 3,
bf16 MFMA tiles).  N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling
over hash tables — every GPU keeps 3 tables (n_hashes = 3·N in total), inputs replicated, ONE
exchange step (all-to-all of the packed per-rank table sums + all-gather of the output).
``value`` = points/s normalised to 3 table passes per point: N_gpus · N_raw / step time.
``--tables-per-gpu 1`` is BASELINE config 4 (n_hashes = #GPUs, one table per GPU).

Prints ONE JSON line on rank 0; see DESIGN.md §6 for every field.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOAD = "tracking-60k"
EVENT_STRIDE = int(os.environ.get("HEPT_BENCH_EVENT_STRIDE", "16"))
TABLES_PER_GPU = 3
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # dense f32-input MFMA


def algorithmic_bytes(n, h, d, c, t, tile_bytes):
    """gather reads of q^,k^,v rows + per-table numer/denom writes, per block_attn launch."""
    e = d + c
    return t * h * n * (2 * e + d) * tile_bytes + t * h * n * (d + 1) * 4


def algorithmic_flops(n, h, d, c, t, b):
    return 2 * t * h * n * b * (2 * d + c)


def cpu_baseline(inp, block_size, min_seconds=10.0):
    """The oracle (CPU restatement of the reference, fp32, all host threads) timed on this box: kind 'port'."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hept_oracle as ho

    cores = os.cpu_count() or 1
    args = [inp[k].cpu() for k in ("q", "k", "v", "coords", "combined_shifts", "w_rpe_weight", "alpha", "out_weight", "out_bias")]
    run = lambda: ho.forward(*args, block_size=block_size, w_per_dist=10, keep=False)["out"]
    # eager CPU torch does not scale to every hardware thread of a big host: give the baseline its
    # best thread count among a few candidates (one forward each), then time that setting
    best = None
    for nt in sorted({min(cores, c) for c in (8, 32, cores)}):
        torch.set_num_threads(nt)
        run()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < min_seconds and len(times) < 12):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": inp["n_raw"] / med, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{len(times)} full forwards of the same {WORKLOAD} cloud (fp32, no_grad, median {med*1e3:.0f} ms; "
                  f"best of 8/32/{cores} threads on a {cores}-thread host)",
    }


@contextlib.contextmanager
def c_stdout_to_stderr():
    """RCCL prints a version banner with C stdio on stdout when a communicator is created; the contract is ONE JSON
    line on stdout.  Route file descriptor 1 to stderr (flushing C stdio on both sides) while RCCL initialises."""
    import ctypes

    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 500 steps = 0.1 s of GPU time; 100-step regions (20 ms) scatter by +-2 % from run to run on one box
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--tables-per-gpu", type=int, default=TABLES_PER_GPU,
                    help="hash tables per GPU (default 3 = BASELINE config 3 at N=1; 1 = config 4: n_hashes = #GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, exchange step, barriers) even with one rank: a self-check")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage HIP-event breakdown to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    # self-check hooks (never set by the driver): HEPT_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL refuses
    # that), HEPT_BENCH_EXCHANGE forces an exchange mode -- together they run the whole N>1 code path on a 1-GPU box
    backend = os.environ.get("HEPT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with c_stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()  # creates the communicator (and prints RCCL's banner) now
            torch.cuda.synchronize()
        group = dist.group.WORLD

    from hept_amd import HEPTAttention, ops
    from hept_amd.synthetic import workload_inputs

    tables_per_gpu = args.tables_per_gpu
    n_tables = tables_per_gpu * world
    inp = workload_inputs(WORKLOAD, seed=0, n_hashes=n_tables)
    B, H, D = 128, 8, 24
    C = inp["coords"].shape[1]
    n, n_raw = inp["q"].shape[0], inp["n_raw"]
    g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}

    attn = HEPTAttention(D + C, h_dim=D, num_heads=H, block_size=B, n_hashes=n_tables, num_w_per_dist=10,
                         precision=args.precision, process_group=group)
    attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                          "e2lsh.alpha": inp["alpha"]}, strict=True)
    attn = attn.to(dev).eval()
    if args.force_dist and world == 1:
        attn.sharding.always_exchange = True
    if multi and os.environ.get("HEPT_BENCH_EXCHANGE"):
        attn.sharding.mode = os.environ["HEPT_BENCH_EXCHANGE"]
    w_rpe = torch.nn.Linear(inp["w_rpe_weight"].shape[1], inp["w_rpe_weight"].shape[0]).to(dev)
    with torch.no_grad():
        w_rpe.weight.copy_(g["w_rpe_weight"])
    kw = dict(w_rpe=w_rpe, coords=g["coords"], combined_shifts=g["combined_shifts"])

    def step():
        with torch.no_grad():
            return attn(g["q"], g["k"], g["v"], **kw)

    def fence():
        torch.cuda.synchronize()
        if multi:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    if multi:
        # first sharded step: if this RCCL build rejects the all-to-all of packed rows, fall back to the reduce-scatter
        # exchange (f32 rows) rather than lose the measurement; the mode used is reported in config.parallelism
        try:
            step()
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print(f"[rank {rank}] all_to_all exchange failed ({exc!r}); using reduce_scatter", file=sys.stderr)
            attn.sharding.mode = "reduce_scatter"
    for _ in range(args.warmup):
        step()
    # HIP events around block_attn on the launch stream, inside the timed region; every 16th step only: an
    # event pair costs ~12 us of stream time per step, sampling keeps `value` within 0.5 % of un-instrumented
    ops.profile_enable(1, args.steps, stride=EVENT_STRIDE)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stage_ms, n_rec = ops.profile_read()
    ops.profile_enable(0)
    if multi:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    ms_per_step = elapsed / args.steps * 1e3

    if args.stages and rank == 0:
        ops.profile_enable(2, args.steps)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        all_ms, cnt = ops.profile_read()
        ops.profile_enable(0)
        print("stage ms/step:", {k: round(v / cnt, 4) for k, v in all_ms.items()}, file=sys.stderr)

    if rank == 0:
        tile_bytes = 2 if args.precision == "bf16" else 4
        attn_ms = stage_ms["block_attn"] / max(n_rec, 1)
        if args.precision == "bf16":
            ach = algorithmic_bytes(n, H, D, C, tables_per_gpu, tile_bytes) / (attn_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        else:
            ach = algorithmic_flops(n, H, D, C, tables_per_gpu, B) / (attn_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF}
        # f32 tiles run as split-bf16 products (6 bf16 MFMAs per f32 product): algorithmic f32 flops against the f32 peak
        roof["kernel"] = "block_attn_kernel" if args.precision == "bf16" else "block_attn_split_kernel"
        roof["kernel_ms"] = attn_ms
        # what an event pair around NOTHING reads on this stream: the bracket's own cost is inside kernel_ms (the
        # rocprofv3 kernel trace in profiles/ shows the kernel itself shorter by about this much)
        gaps = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            torch.cuda.synchronize()
            gaps.append(e0.elapsed_time(e1))
        roof["event_pair_overhead_ms"] = sorted(gaps)[len(gaps) // 2]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "attn_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.precision)
            except Exception:
                traffic = None
        roof["traffic"] = traffic
        line = {
            "metric": "attention-fwd points/sec", "value": world * n_raw / (elapsed / args.steps), "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD}: N_raw={n_raw} padded N={n}, block_size={B}, n_hashes={tables_per_gpu}/GPU "
                                   f"({n_tables} total), H={H}, D={D}, C={C}, tiles {args.precision}",
                       "parallelism": f"tables sharded {tables_per_gpu}/GPU over {world} GPU(s)" +
                                      (f", exchange {attn.sharding.mode}" if multi else ""),
                       "hbm_algorithmic_GBps_block_attn": algorithmic_bytes(n, H, D, C, tables_per_gpu, tile_bytes) / (attn_ms * 1e-3) / 1e9},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(inp, B)
        print(json.dumps(line), flush=True)
    if multi:
        with c_stdout_to_stderr():
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
