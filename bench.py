#!/usr/bin/env python3
"""bench.py — HEPT attention-forward throughput on MI355X (BASELINE.json metric).

One "step" = one ``HEPTAttention.forward`` (the whole hot path: E2LSH hash -> sort -> block
attention -> combine + out_linear) over one synthetic tracking-60k cloud whose inputs are already
resident in HBM.  N = 1: BASELINE config 3 (N_raw 60000 -> 60032 padded, block 128, n_hashes 3,
bf16 MFMA tiles).  N > 1: weak scaling over hash tables — every GPU keeps 3 tables (n_hashes = 3·N
in total), inputs replicated, one exchange (all-to-all of the packed per-rank table sums, pipelined
by head groups behind the block attention, + all-gather of the output).
``value`` = points/s normalised to 3 table passes per point: N_gpus · N_raw / step time.

Launching: ``python bench.py --gpus N`` works on its own — with ``WORLD_SIZE`` unset and N > 1 this
process starts N children (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment)
*before it makes any GPU call*, never touches the GPU itself, relays rank 0's JSON line and exits
non-zero if any child failed.  Under ``torch.distributed.run`` (WORLD_SIZE set) it is a rank.

Beside the headline the line carries: ``fp32`` (N = 1: the reference-precision run of the same
workload, measured in the same process before the headline region), ``c4`` (BASELINE config 4: one table
per GPU, n_hashes = N), ``roofline`` (HBM bound, dominant kernel, HIP events on the launch stream)
and ``cpu_baseline`` (N = 1).  See DESIGN.md §6 for every field.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP IPC between the ranks of a node (RCCL's and the one-sided exchange's) needs the dmabuf mode on this stack; set
# before anything initialises the GPU, also when a launcher started this process as a rank
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOAD = "tracking-60k"
TABLES_PER_GPU = 3
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec ...
HBM_COPY_GBS = 6290.0      # ... and the 6.29 TB/s the same guide measures for a device copy (roofline.frac_of_copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA (both the bf16 and the split-bf16 f32 kernel issue bf16 MFMAs)
B, H, D = 128, 8, 24


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 500 steps = 0.1 s of GPU time; 100-step regions (20 ms) scatter by +-2 % from run to run on one box
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--tables-per-gpu", type=int, default=TABLES_PER_GPU,
                    help="hash tables per GPU (default 3 = BASELINE config 3 at N=1; 1 = config 4: n_hashes = #GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the fp32 and c4 sub-records")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, exchange step, barriers) even with one rank: a self-check")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage HIP-event breakdown to stderr")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher (no GPU)
def launch(args) -> int:
    """Start one child per GPU and relay rank 0's line.  Nothing here may touch the GPU: a process that has
    initialised HIP must not be the parent of the ranks' device contexts (and must never exec)."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), HEPT_BENCH_CHILD="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the JSON line; the other ranks' stdout goes to stderr so that stdout stays one line
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    out0 = b""
    failed = None
    try:
        alive = set(range(n))
        # read rank 0's pipe without blocking the watch on the others
        os.set_blocking(procs[0].stdout.fileno(), False)
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is not None:
                    alive.discard(r)
                    if rc != 0 and failed is None:
                        failed = (r, rc)
            try:
                chunk = procs[0].stdout.read()
                if chunk:
                    out0 += chunk
            except (BlockingIOError, ValueError):
                pass
            if failed is not None:
                break
            time.sleep(0.02)
    finally:
        for p in procs:  # exact PIDs only
            if p.poll() is None:
                if failed is not None:
                    p.terminate()
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    try:
        rest = procs[0].stdout.read()
        if rest:
            out0 += rest
    except (BlockingIOError, ValueError):
        pass
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        sys.stderr.write(out0.decode(errors="replace"))
        return failed[1] if failed[1] > 0 else 1
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------ rank body
def algorithmic_bytes(n, h, d, c, t, tile_bytes):
    """gather reads of q^,k^,v rows + per-table numer/denom writes, per block_attn launch."""
    e = d + c
    return t * h * n * (2 * e + d) * tile_bytes + t * h * n * (d + 1) * 4


def algorithmic_flops(n, h, d, c, t, b):
    return 2 * t * h * n * b * (2 * d + c)


SMALL_SORT_MAX = 6144      # csrc/sort_tables.hip SMALL_CAP: clouds up to this size sort in one launch
RIDERS_MAX = 131072        # ... and up to this size the bucket-sort launch can carry the v rows (hept_sort_carries_rows)


def step_kernels(n, h, d, c, t, precision, block_size=128):
    """Algorithmic bytes of every launch of one hept_forward: what each kernel has to read and
    write once, whatever the caches do.  Returns [(stage key, kernel name, bytes)] in launch order; stage keys are those of
    ops.profile_read()."""
    s = 4 if precision == "fp32" else 2
    row = 32 * s                                              # one q^ / k^ / v row
    keys = 2 * t * h * n                                      # sort keys: q and k segments
    v_rows = n * h * d * 4 + n * h * row                      # v in, v halves of the kvhat rows out
    # csrc/capi.hip run_begin: f32 rows above block 128 read the values in place (direct_v_pays): nobody builds v rows
    direct_v = precision == "fp32" and block_size > 128 and d % 4 == 0 and not os.environ.get("HEPT_NO_DIRECT_V")
    riders = ((t >= 2 or precision == "fp32") and SMALL_SORT_MAX < n <= RIDERS_MAX and not direct_v
              and not os.environ.get("HEPT_NO_ROW_RIDERS")) or (os.environ.get("HEPT_FORCE_ROW_RIDERS") and not direct_v)
    if direct_v:
        v_rows = 0
    prep = 2 * n * h * d * 4 + 2 * n * c * 4 + t * h * n * 8 // 8 + 2 * n * h * row + keys * 4
    packed = precision != "fp32" and d == 24
    combine = t * n * h * (64 if packed else 128) + n * d * 4
    attn = algorithmic_bytes(n, h, d, c, t, s)
    attn_name = "block_attn_split_kernel" if precision == "fp32" else "block_attn_kernel"
    if n <= SMALL_SORT_MAX:
        return [("prep_hash", "prep_hash_kernel (q, k, v roles)", prep + v_rows),
                ("sort_tables", "small_sort_kernel", keys * 4 + t * h * n * 8 + keys * 4),
                ("block_attn", attn_name, attn), ("combine", "combine_out_kernel", combine)]
    chunk = keys * 4 + t * h * n * 8 + keys * 8               # hashes, int64 codes (once per (table, head)), pairs out
    bucket = keys * 8 + keys * 4                               # pairs in, positions out
    return [("prep_hash", "prep_hash_kernel (q, k roles)" if riders else "prep_hash_kernel (q, k, v roles)",
             prep + (0 if riders else v_rows)),
            ("chunk_sort", "chunk_sort_kernel", chunk),
            ("bucket_sort", "bucket_sort_kernel + v-row riders" if riders else "bucket_sort_kernel", bucket + (v_rows if riders else 0)),
            ("block_attn", attn_name, attn), ("combine", "combine_out_kernel", combine)]


def cpu_baseline(inp, block_size, min_seconds=10.0):
    """The oracle (CPU restatement of the reference, fp32, all host threads) timed on this box: kind 'port'."""
    import torch

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
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": inp["n_raw"] / med, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
        "cpu_model": model, "host_threads": cores,
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


def attn_source_sha256():
    """Digest of the block-attention sources of THIS tree (tools/make_traffic.py stores the same digest with a PMC pass)"""
    import hashlib

    h = hashlib.sha256()
    for rel in ("hept_amd/csrc/block_attn.hip", "hept_amd/csrc/common.h", "hept_amd/csrc/p2p_dev.h", "hept_amd/csrc/Makefile"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def launched_kernel(precision, block_size=128, head_dim=24):
    """The template instance hept_block_attn launches for this workload (csrc/block_attn.hip: block_attn_impl)"""
    nkt, full = (block_size + 31) // 32, "true" if block_size % 32 == 0 else "false"
    if precision == "fp32":
        return f"block_attn_split_kernel<{nkt},{full},3,false>"
    p16 = "true" if head_dim == 24 else "false"
    return f"block_attn_kernel<{nkt},true,{p16},{'true' if precision == 'mixed16' else 'false'},{full},false>"


def pmc_record(record_key, precision, block_size=128, head_dim=24, write_bytes=None):
    """Per-launch PMC figures of the block-attention kernel from profiles/attn_traffic.json (tools/pmc_all.sh passes):
    L2 memory-side traffic in bytes, the fraction of the kernel's cycles the matrix pipe was busy and the fraction the
    vector ALUs were issuing -- counters cannot be collected inside an un-profiled run, so they are copied, but ONLY
    when the record belongs to this build AND this workload: same digest of the kernel sources, the record key of
    this sub-record (``c3/bf16`` ...), the same template instance as the one this run launches, and -- the check that
    would have caught round 5's mix-up -- WRITE_SIZE within 1 % of the bytes this launch has to write
    (``write_bytes``: T*H*N partial rows).  Returns (record dict or None, source)."""
    tpath = os.path.join(ROOT, "profiles", "attn_traffic.json")
    key = launched_kernel(precision, block_size, head_dim)
    wkey = f"{record_key}/{precision}"
    src = {"file": "profiles/attn_traffic.json", "workload_key": wkey, "kernel_launched": key}
    try:
        rec = json.load(open(tpath))
    except Exception as exc:  # noqa: BLE001
        src["refused"] = f"unreadable: {exc!r}"
        return None, src
    src.update(git_head=rec.get("git_head"), source_sha256=rec.get("source_sha256"))
    if rec.get("source_sha256") != attn_source_sha256():
        src["refused"] = "the record's source digest is not this tree's (re-run tools/pmc_all.sh)"
        return None, src
    ent = ((rec.get("by_workload") or {}).get(wkey) or {}).get(key)
    if not ent:
        src["refused"] = "no PMC pass for this (workload, kernel template)"
        return None, src
    if write_bytes is not None and abs(ent.get("write_bytes", 0.0) - write_bytes) > 0.01 * write_bytes:
        src["refused"] = (f"the record's WRITE_SIZE ({ent.get('write_bytes', 0.0):.0f} B) is not this launch's partial rows "
                          f"({write_bytes} B): another workload's pass")
        return None, src
    src.update(kernel_measured=ent.get("kernel"), command=ent.get("command"))
    return ent, src


def bound_from_counters(ent, kernel_ms):
    """Which resource the kernel sits closest to, from its counters and this run's duration: HBM (traffic over the
    measured copy rate) against instruction issue (vector ALU + matrix pipe shares of the kernel's cycles -- the two
    share a wave's issue slot).  The roofline contract knows two bounds: issue-bound is reported as "mfma"."""
    hbm = ent["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_GBS
    issue = (ent.get("valu_issue_frac") or 0.0) + (ent.get("mfma_busy_frac") or 0.0)
    return ("hbm" if hbm >= issue else "mfma"), {"hbm_frac_of_copy_on_traffic": hbm, "valu_issue_frac": ent.get("valu_issue_frac"),
                                                  "mfma_busy_frac": ent.get("mfma_busy_frac"), "issue_frac": issue}


def worker(args) -> int:
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("HEPT_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        raise SystemExit(f"rank {rank}: HEPT_BENCH_FAIL_RANK set")
    # self-check hooks (never set by the driver): HEPT_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL refuses
    # that), HEPT_BENCH_EXCHANGE forces an exchange mode -- together they run the whole N>1 code path on a 1-GPU box
    backend = os.environ.get("HEPT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: hept_amd has no CPU path")
    if backend == "nccl" and n_dev < world:
        raise SystemExit(f"--gpus {world} but only {n_dev} GPU(s) are visible (RCCL needs one device per rank)")
    dev_index = local_rank % n_dev if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    multi = world > 1 or args.force_dist
    dist = None
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
    from hept_amd.synthetic import WORKLOADS, workload_inputs

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    host_inputs = []   # kept to the end: returning ~150 MB of host pages to the OS takes the CPU ~17 ms, which is
                       # 17 ms of idle device right in front of the next timed region if it happens between two

    def build(tables_per_gpu, precision, workload=WORKLOAD, block_size=None):
        """Module + resident inputs of one named workload (hept_amd.synthetic.WORKLOADS = the BASELINE.json configs);
        ``tables_per_gpu`` None: the workload's own n_hashes; ``block_size``: override (the reference's yaml uses 100)"""
        over = {}
        if tables_per_gpu is not None:
            over["n_hashes"] = tables_per_gpu * world
        if block_size is not None:
            over["block_size"] = block_size
        inp = workload_inputs(workload, seed=0, **over)
        host_inputs.append(inp)
        g = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
        c = inp["coords"].shape[1]
        n_tables = inp["alpha"].shape[2]
        bs = block_size if block_size is not None else WORKLOADS[workload]["block_size"]
        attn = HEPTAttention(D + c, h_dim=D, num_heads=H, block_size=bs, n_hashes=n_tables, num_w_per_dist=10,
                             precision=precision, process_group=group)
        attn.load_state_dict({"out_linear.weight": inp["out_weight"], "out_linear.bias": inp["out_bias"],
                              "e2lsh.alpha": inp["alpha"]}, strict=True)
        attn = attn.to(dev).eval()
        attn.reserve(inp["q"].shape[0], c, dev)   # workspace now, not inside the first warm-up step
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

        return inp, attn, step

    def launches_per_step(attn):
        """block_attn launches per forward: the pipelined exchange runs the block attention one head group at a time"""
        sh = attn.sharding
        if sh is not None and sh.mode == "all_to_all" and (sh.world > 1 or sh.always_exchange):
            return sh.groups_for(H)
        return 1

    NO_SAMPLES = 1 << 30   # a stride no run reaches: the pool stays allocated, nothing is bracketed
    # (several ranks: up to HEPT_MAX_HEAD_GROUPS = 8 block_attn launches per step)
    sub_cap = 200   # steps of a sub-record
    ops.profile_enable(1, (max(args.steps, sub_cap) + 2) * (8 if multi else 1), stride=NO_SAMPLES)

    def measure(step, steps, warmup, launches=1):
        """W untimed steps, then exactly K steps between fences; HIP events around block_attn on the launch stream for
        a sample of the steps (an event pair costs stream time, so the stride keeps >= 80 % of the steps bare).
        Returns (elapsed s, mean block_attn ms per step, event samples)."""
        # 4 ... 32 bracketed steps: an event pair is a barrier packet on either side of the kernel (~4 us of bubble
        # on this stack, tools/event_cost.py), so a 20-step region keeps 16 of its steps bare.  The event pool is the
        # process-wide one made at start-up: creating or destroying hundreds of events takes the host milliseconds,
        # and a device left idle that long starts the next region below its steady clocks.
        stride = max(1, steps // max(4, min(32, steps // 5)))
        call_stride = stride * launches + (1 if launches > 1 else 0)
        ops.profile_stride(call_stride)
        for _ in range(warmup):
            step()
        ops.profile_read()             # drops the warm-up samples
        ops.profile_stride(call_stride)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        stage_ms, n_rec = ops.profile_read()
        ops.profile_stride(NO_SAMPLES)
        if multi:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt)
        return elapsed, stage_ms["block_attn"] / max(n_rec, 1) * launches, n_rec

    def measure_median(step, steps, warmup, regions=3):
        """A sub-record's region on a device that has just idled through a module build (tens of ms of host work) can
        fall inside its clock ramp, and one scheduling hiccup of a shared host doubles a 50 ms region (one refresh of round 6
        read 431 us per forward for the fp32 record, 264-275 us in every other run): keep the device busy for 30 ms, then
        take the median of three regions -- as the per-configuration records do."""
        t_busy = time.perf_counter()
        while time.perf_counter() - t_busy < 0.03:
            step()
        runs = sorted((measure(step, steps, warmup) for _ in range(regions)), key=lambda r: r[0])
        return runs[len(runs) // 2]

    def roofline(n, c, tables, precision, attn_ms, n_rec, block_size=B, record_key="c3"):
        """HBM roofline of the block-attention kernel: algorithmic bytes per launch / mean launch duration.  The f32
        kernel issues bf16 MFMAs (split products); ``bound`` is read off the kernel's counters (profiles/attn_traffic.json,
        one PMC record per kernel template: HBM traffic against vector + matrix issue) when a record of this build
        exists, and is "hbm" with ``traffic`` null otherwise."""
        tile_bytes = 4 if precision == "fp32" else 2
        nbytes = algorithmic_bytes(n, H, D, c, tables, tile_bytes)
        # the event pair's own cost sits inside what it brackets: half of a lone pair's reading is taken off, as for the
        # `kernels` list (whole_step) -- with it the duration agrees with the rocprofv3 kernel trace of profiles/ within
        # 1-2 % (raw: 2-5 % above it); the raw reading stays in the record
        attn_raw_ms, attn_ms = attn_ms, max(attn_ms - event_pair_ms() / 2, 1e-6)
        ach = nbytes / (attn_ms * 1e-3) / 1e9
        row_bytes = 64 if (precision != "fp32" and D == 24) else 128           # packed / f32 partial rows
        ent, source = pmc_record(record_key, precision, block_size, write_bytes=tables * H * n * row_bytes)
        bound, evidence = ("hbm", None) if ent is None else bound_from_counters(ent, attn_ms)
        flops = algorithmic_flops(n, H, D, c, tables, block_size)
        # `frac` is on the RAW event reading (the contract's live measurement); `frac_adjusted` takes half of a lone event
        # pair's cost off it (both, and the method named)
        ach_raw = nbytes / (attn_raw_ms * 1e-3) / 1e9
        return {"bound": bound, "bound_evidence": evidence, "achieved": ach_raw, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_raw / HBM_PEAK_GBS, "frac_adjusted": ach / HBM_PEAK_GBS, "frac_of_copy": ach_raw / HBM_COPY_GBS,
                "copy_peak": HBM_COPY_GBS,
                "timing_method": "HIP events on the launch stream around the kernel (kernel_ms_raw); kernel_ms / frac_adjusted: "
                                 "minus half of a lone event pair's cost, which is what agrees with the rocprofv3 kernel trace",
                "kernel": "block_attn_kernel" if precision != "fp32" else "block_attn_split_kernel",
                "kernel_ms": attn_ms, "kernel_ms_raw": attn_raw_ms, "event_samples": n_rec, "algorithmic_bytes": nbytes,
                "traffic": ent["traffic"] if ent else None, "traffic_source": source,
                "mfma_busy_frac": ent.get("mfma_busy_frac") if ent else None,
                "algorithmic_tflops": flops / (attn_ms * 1e-3) / 1e12, "mfma_bf16_peak_tflops": MFMA_BF16_PEAK_TF}

    _pair_cost = []

    def event_pair_ms():
        """what an event pair around NOTHING reads on this stream (median of 20): the bracket's own cost is inside every
        event-timed stage (the rocprofv3 kernel trace in profiles/ shows the kernels shorter by about this much)"""
        if not _pair_cost:
            gaps = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e1.record()
                torch.cuda.synchronize()
                gaps.append(e0.elapsed_time(e1))
            _pair_cost.append(sorted(gaps)[len(gaps) // 2])
        return _pair_cost[0]

    def whole_step(step, n, c, tables, precision, ms_per_step, steps=60):
        """Every launch of the step, not only the dominant one: an extra, untimed pass with all stages bracketed by HIP
        events (hept_profile mode 2), the bracket's own cost taken off each stage, against the launch's algorithmic
        bytes -- and the whole step's bytes over the TIMED step time (``step_roofline``)."""
        ops.profile_enable(2, steps + 2)
        fence()
        for _ in range(steps):
            step()
        fence()
        ms, cnt = ops.profile_read()
        ops.profile_enable(1, (max(args.steps, sub_cap) + 2) * (8 if multi else 1), stride=NO_SAMPLES)
        if not cnt:
            return {}
        per = {k: v / cnt for k, v in ms.items()}
        per["bucket_sort"] = per["sort_tables"] - per["chunk_sort"]
        pair = event_pair_ms()
        kernels, total = [], 0
        for key, name, nbytes in step_kernels(n, H, D, c, tables, precision):
            # (a chain of stages shares its events: every stage is bounded by two of them but each event bounds two stages,
            #  so a stage carries about half of what a lone pair costs -- with this the five launches add up to the timed,
            #  un-bracketed step within 1 %)
            k_ms = max(per[key] - pair / 2, 1e-6)
            gbs = nbytes / (k_ms * 1e-3) / 1e9
            kernels.append({"name": name, "ms": k_ms, "algorithmic_bytes": nbytes, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
            total += nbytes
        gbs = total / (ms_per_step * 1e-3) / 1e9
        return {"kernels": kernels, "kernels_note": f"HIP-event stage times of {cnt} extra untimed steps minus half of a lone event pair's {pair * 1e3:.1f} us",
                "step_roofline": {"bound": "hbm", "algorithmic_bytes": total, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gbs / HBM_PEAK_GBS, "frac_of_copy": gbs / HBM_COPY_GBS}}

    def exchange_record(attn, step, steps=30):
        """What ran between the GPUs, said by the objects that ran it: the transport and head groups after the ladder
        and tuning, the world size of the communicator that moved the rows (hept_comm_world -- not torch's process
        group), and per-term HIP-event times of the sharded step (one extra, untimed pass with every stage bracketed:
        exposed push / transfer of the last head group, combine, output gather)."""
        sh = attn.sharding
        rec = {"describe": sh.describe(), "head_groups": launches_per_step(attn), "comm_world": 0, "transport": "torch.distributed"}
        if sh._native:
            from hept_amd import _lib

            rec["comm_world"] = int(_lib.load().hept_comm_world(sh._native))
            rec["transport"] = "rccl" if (sh._p2p_failed or sh.exchange == "rccl") else "one-sided"
            ops.profile_enable(2, steps + 2)
            fence()
            for _ in range(steps):
                step()
            fence()
            ms, cnt = ops.profile_read()
            ops.profile_enable(1, (max(args.steps, sub_cap) + 2) * (8 if multi else 1), stride=NO_SAMPLES)
            if cnt:
                us = {k: v / cnt * 1e3 for k, v in ms.items()}
                rec.update(samples=cnt, prep_us=us["prep_hash"], sort_us=us["sort_tables"], block_attn_us=us["block_attn"],
                           exposed_push_us=us["combine"], combine_us=us["sharded_combine"], gather_us=us["sharded_gather"])
        return rec

    def exchange_proxy(precision, steps=100):
        """BASELINE config 4's per-rank shape (one table) on a ONE-rank communicator with the exchange forced on, per
        transport: what the sharded step costs in launches, local copies and stores into the (uncached) exchange buffers
        before any xGMI traffic -- the terms of the exchange a single GPU can measure, so that the line
        shows which transport tune() would be choosing between.  us per step."""
        import torch.distributed as dist
        from hept_amd.sharding import TableSharding

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        rec = {}
        with c_stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
            dist.barrier()
            torch.cuda.synchronize()
        try:
            inp1 = workload_inputs(WORKLOAD, seed=0, n_hashes=1)
            host_inputs.append(inp1)
            g1 = {k: v.to(dev) for k, v in inp1.items() if torch.is_tensor(v)}
            c1 = inp1["coords"].shape[1]
            w_rpe1 = torch.nn.Linear(inp1["w_rpe_weight"].shape[1], inp1["w_rpe_weight"].shape[0]).to(dev)
            with torch.no_grad():
                w_rpe1.weight.copy_(g1["w_rpe_weight"])
            kw1 = dict(w_rpe=w_rpe1, coords=g1["coords"], combined_shifts=g1["combined_shifts"])
            # one_sided_view: TableSharding(out_view=True) -- the step ends at the last output flag, the caller reads the
            # gathered output in the exchange buffer (no copy out of it)
            for name, mode, via in (("plain", None, None), ("one_sided", "all_to_all", "p2p"),
                                    ("one_sided_view", "all_to_all", "p2p"), ("rccl", "all_to_all", "rccl"),
                                    ("all_reduce", "all_reduce", None)):
                m = HEPTAttention(D + c1, h_dim=D, num_heads=H, block_size=B, n_hashes=1, num_w_per_dist=10, precision=precision,
                                  process_group=dist.group.WORLD if mode else None)
                if mode:
                    m.sharding = TableSharding(1, dist.group.WORLD, mode=mode, always_exchange=True, head_groups=1,
                                               out_view=name.endswith("_view"))
                    if via:
                        m.sharding.exchange = via
                m.load_state_dict({"out_linear.weight": inp1["out_weight"], "out_linear.bias": inp1["out_bias"],
                                   "e2lsh.alpha": inp1["alpha"]}, strict=True)
                m = m.to(dev).eval()
                with torch.no_grad(), c_stdout_to_stderr():
                    for _ in range(10):
                        m(g1["q"], g1["k"], g1["v"], **kw1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        m(g1["q"], g1["k"], g1["v"], **kw1)
                    torch.cuda.synchronize()
                    rec[name] = round((time.perf_counter() - t0) / steps * 1e6, 1)
                if mode:
                    m.sharding.check()
                del m
        except Exception as exc:  # noqa: BLE001  (a record beside the contract's line: never fail the run for it)
            rec["error"] = repr(exc)
        finally:
            with c_stdout_to_stderr():
                dist.destroy_process_group()
        return rec

    tables_per_gpu = args.tables_per_gpu
    inp, attn, step = build(tables_per_gpu, args.precision)
    n, n_raw, C = inp["q"].shape[0], inp["n_raw"], inp["coords"].shape[1]
    def settle(attn, step):
        """First sharded steps: walk down the transport ladder (one-sided xGMI stores -> RCCL from the C library ->
        torch.distributed -> reduce-scatter) until one works on EVERY rank -- a rank-local failure must not leave the
        others inside an exchange of a different kind, so the ranks agree through a MAX over a failure flag."""
        dist.barrier()
        while True:
            failed = 0
            try:
                step()
                step()
                torch.cuda.synchronize()
                attn.sharding.check()
            except Exception as exc:  # noqa: BLE001
                print(f"[rank {rank}] exchange '{attn.sharding.describe()}' failed: {exc!r}", file=sys.stderr)
                failed = 1
            flag = torch.tensor([failed], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag) == 0:
                break
            if not attn.sharding.downgrade():
                raise SystemExit("no working exchange left")
        # which transport and how many head groups are fastest is a property of the machine (xGMI store rate against
        # RCCL's launch latency): measure, keep the best
        table = attn.sharding.tune(step, dev)
        if rank == 0 and table:
            print("exchange tuning (us/step): " + ", ".join(f"{t}/{g}: {v * 1e6:.1f}" for (t, g), v in sorted(table.items())),
                  file=sys.stderr)
        return table

    def tune_record(table):
        """tune()'s table for the JSON line: us per step of every (transport, head groups) candidate that worked on all
        ranks, and what RCCL alone would have done (the best RCCL candidate) next to the one-sided transport"""
        if not table:
            return {}
        rec = {"tune_us_per_step": {f"{t}/{g}": round(v * 1e6, 1) for (t, g), v in sorted(table.items())}}
        rccl = [v for (t, _), v in table.items() if t == "rccl"]
        p2p = [v for (t, _), v in table.items() if t == "p2p"]
        rec["rccl_only_us_per_step"] = round(min(rccl) * 1e6, 1) if rccl else None
        rec["one_sided_us_per_step"] = round(min(p2p) * 1e6, 1) if p2p else None
        return rec

    head_tune = {}
    if multi:
        head_tune = tune_record(settle(attn, step))

    # ------------------------------------------------------------------------------------------------- sub-records
    def sub_records():
        """The f32-tile record (N = 1) and BASELINE config 4 (one table per GPU) on the same workload"""
        sub = {}
        if not args.no_extra:
            # one process: long enough (60 + 25 ms of device time) to leave the device in its steady state
            sub_steps = sub_cap if not multi else max(20, min(args.steps, sub_cap))
            sub_warm = max(3, min(args.warmup, 10))
            if world == 1 and not args.force_dist and args.precision == "bf16":
                # reference precision (f32 tiles) on the same workload, same process
                _, attn32, step32 = build(tables_per_gpu, "fp32")
                el, ams, nrec = measure_median(step32, sub_steps, sub_warm)
                sub["fp32"] = {"ms_per_step": el / sub_steps * 1e3, "value": n_raw / (el / sub_steps), "unit": "points/s",
                               "steps": sub_steps,
                               "dtype": "f32 rows and accumulation; tile products as split-bf16 MFMAs (q^.k^: 6 terms, P.V: P in two bf16 "
                                        "pieces = 16 significand bits); precision='fp32_mfma' is the exact f32 mode",
                               "roofline": roofline(n, C, tables_per_gpu, "fp32", ams, nrec)}
                sub["fp32"].update(whole_step(step32, n, C, tables_per_gpu, "fp32", el / sub_steps * 1e3))
                del attn32, step32
                # mixed16 (fp16 q^/k^ rows, bf16 weights and values): the 16-bit mode whose EVERY row stays within
                # 2.5e-2 of the fp32 reference's row scale (bf16: 1e-1 on trained weights), at the same speed
                _, attn16, step16 = build(tables_per_gpu, "mixed16")
                el, ams, nrec = measure_median(step16, sub_steps, sub_warm)
                sub["mixed16"] = {"ms_per_step": el / sub_steps * 1e3, "value": n_raw / (el / sub_steps), "unit": "points/s",
                                  "steps": sub_steps, "dtype": "f16 q^,k^ rows / bf16 weights, values",
                                  "block_attn_ms": ams}
                del attn16, step16
            if world == 1 and not args.force_dist and args.precision == "bf16":
                # the other BASELINE.json configurations on this GPU, each in both tile precisions: c1 (example-4k),
                # c2 (tracking-6k), c5 (pileup batch, block 256) and b100 = the headline cloud at the reference's own
                # block_size 100 (src/configs/tracking/tracking_trans_hept.yaml:12).  Short clouds are latency-bound:
                # their roofline fraction says how far a 40 us forward is from streaming its bytes.
                # c2x10: ten tracking-6k clouds in ONE call through the batch index of the AND code -- the way the
                # reference runs small clouds (example/transformer.py:35-63), at the per-point rate of the 60k cloud
                for key, wl, bs in (("c1", "example-4k", None), ("c2", "tracking-6k", None), ("c2x10", "tracking-6k-x10", None),
                                    ("c5", "pileup-8clouds", None), ("b100", WORKLOAD, 100)):
                    rec = {"workload": wl + (f" at block_size={bs}" if bs else "")}
                    for prec in ("fp32", "bf16"):
                        inp_s, attn_s, step_s = build(None, prec, wl, bs)
                        bsz = bs if bs is not None else WORKLOADS[wl]["block_size"]
                        ns, cs, ts = inp_s["q"].shape[0], inp_s["coords"].shape[1], inp_s["alpha"].shape[2]
                        # short clouds are bound by the host's launch rate (33 us of issue per forward), and a 10 ms
                        # region is at the mercy of one scheduling hiccup of a shared host: median of three regions
                        # (three regions for the long clouds as well: one hiccup -- an allocator call of the module built
                        #  before, a scheduling pause of the host -- doubled a 40 ms region in one run of round 5)
                        # ... and the device idled while build() made this record's inputs on the host: a short cloud's three
                        # regions (7 ms each) can all fall inside its clock ramp (one refresh of round 5 read 98 us per
                        # forward for example-4k that way, 35 us in the runs before and after) -- keep it busy for 30 ms first
                        el, ams, nrec = measure_median(step_s, sub_steps, sub_warm)
                        roof_s = roofline(ns, cs, ts, prec, ams, nrec, bsz, record_key=key)
                        rec[prec] = {"ms_per_step": el / sub_steps * 1e3, "value": inp_s["n_raw"] / (el / sub_steps),
                                     "unit": "points/s", "steps": sub_steps, "n_raw": inp_s["n_raw"], "n_padded": ns,
                                     "block_size": bsz, "n_hashes": ts, "regions": 3,
                                     "roofline": {k: roof_s[k] for k in ("bound", "bound_evidence", "achieved", "peak", "unit", "frac",
                                                                         "kernel", "kernel_ms", "algorithmic_bytes", "traffic",
                                                                         "mfma_busy_frac")}}
                        rec[prec]["roofline"]["kernel_launched"] = roof_s["traffic_source"]["kernel_launched"]
                        if roof_s["traffic"] is None:
                            rec[prec]["roofline"]["traffic_refused"] = roof_s["traffic_source"].get("refused")
                        del attn_s, step_s
                    sub[key] = rec
            if tables_per_gpu != 1:
                # BASELINE config 4: n_hashes = #GPUs, one table per GPU (at N = 1 a single table)
                _, attn4, step4 = build(1, args.precision)
                c4_tune = {}
                if multi:
                    c4_tune = tune_record(settle(attn4, step4))
                if multi:
                    el, ams, nrec = measure(step4, sub_steps, sub_warm, launches_per_step(attn4))
                else:   # (one process: a median of three regions, like every other sub-record)
                    el, ams, nrec = measure_median(step4, sub_steps, sub_warm)
                if multi:
                    attn4.sharding.check()
                sub["c4"] = {"workload": f"{WORKLOAD}, n_hashes={world} sharded 1 per GPU over {world} GPU(s)",
                             "ms_per_step": el / sub_steps * 1e3, "value": world * n_raw / (el / sub_steps),
                             "unit": "points/s (N_gpus * N_raw / step time, one table pass per point and GPU)",
                             "steps": sub_steps, "block_attn_ms": ams}
                if multi:
                    sub["c4"]["exchange"] = exchange_record(attn4, step4)
                    sub["c4"]["exchange"].update(c4_tune)
                    attn4.sharding.check()
                del attn4, step4
        return sub

    # One process: the sub-records run BEFORE the headline region -- they are part of every default run anyway, and
    # after them the device is in its steady state (clocks up, code objects and workspaces resident), which W = 5
    # steps (1 ms) alone do not reach on a GPU that idled while the process started (tools/event_cost.py, DESIGN.md
    # section 6).  Several ranks: settle() has already run hundreds of steps, and the headline module's exchange
    # buffers are released before config 4 builds its own.
    sub = sub_records() if not multi else {}

    # ---------------------------------------------------------------------------------------------- headline region
    elapsed, attn_ms, n_rec = measure(step, args.steps, args.warmup, launches_per_step(attn))
    ms_per_step = elapsed / args.steps * 1e3
    exch = None
    if multi:
        attn.sharding.check()   # a one-sided wait that timed out inside the region voids the measurement
        exch = exchange_record(attn, step)
        exch.update(head_tune)
        attn.sharding.check()

    if args.stages and rank == 0:
        ops.profile_enable(2, args.steps)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        all_ms, cnt = ops.profile_read()
        ops.profile_enable(1, (max(args.steps, sub_cap) + 2) * (8 if multi else 1), stride=NO_SAMPLES)
        print("stage ms/step:", {k: round(v / cnt, 4) for k, v in all_ms.items()}, file=sys.stderr)

    line = None
    if rank == 0:
        roof = roofline(n, C, tables_per_gpu, args.precision, attn_ms, n_rec)
        roof["event_pair_overhead_ms"] = event_pair_ms()
        n_tables = tables_per_gpu * world
        line = {
            "metric": "attention-fwd points/sec", "value": world * n_raw / (elapsed / args.steps), "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32 (split-bf16 tile products)", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD}: N_raw={n_raw} padded N={n}, block_size={B}, n_hashes={tables_per_gpu}/GPU "
                                   f"({n_tables} total), H={H}, D={D}, C={C}, tiles {args.precision}",
                       "parallelism": f"tables sharded {tables_per_gpu}/GPU over {world} GPU(s)" +
                                      (f", exchange {attn.sharding.describe()}" if multi else ""),
                       # ranks of torch's RCCL process group (rendezvous, barriers) ...
                       "rccl_ranks": (dist.get_world_size() if multi and backend == "nccl" else 0),
                       # ... and of the communicator that actually moved the rows (0: torch.distributed collectives)
                       "comm_world": exch["comm_world"] if exch else 0,
                       "transport": exch["transport"] if exch else None,
                       "hbm_algorithmic_GBps_block_attn": roof["achieved"]},
            "roofline": roof,
        }
        if exch:
            line["exchange"] = exch
    if not multi:
        ws = whole_step(step, n, C, tables_per_gpu, args.precision, ms_per_step)
        if rank == 0:
            line.update(ws)
    del attn, step
    if multi:
        sub = sub_records()
    if rank == 0:
        line.update(sub)

    if rank == 0 and world == 1 and not multi and not args.no_extra and "c4" in line:
        line["c4"]["exchange_on_one_rank_us"] = exchange_proxy(args.precision)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(inp, B)
        # The driver keeps the last 8 KB of stdout: the long per-configuration sub-records go to the FRONT of the line, the
        # contract's fields, the headline roofline, the CPU baseline, the reference-precision record and the per-launch
        # list to its END (round 5 lost `fp32`, `kernels` and `step_roofline` off the front of that tail), closed by a
        # short `summary` that repeats the numbers a reader looks for first.
        front = [k for k in ("c1", "c2", "c2x10", "c5", "b100", "c4", "mixed16", "exchange") if k in line]
        back = [k for k in ("cpu_baseline", "fp32", "roofline", "kernels", "kernels_note", "step_roofline") if k in line]
        ordered = {k: line[k] for k in front}
        ordered.update({k: v for k, v in line.items() if k not in front and k not in back})
        ordered.update({k: line[k] for k in back})
        summ = {"ms_per_step": line["ms_per_step"], "points_per_s": line["value"],
                "block_attn_ms": line["roofline"]["kernel_ms"], "roofline_frac": line["roofline"]["frac"],
                "traffic": line["roofline"]["traffic"], "mfma_busy_frac": line["roofline"]["mfma_busy_frac"]}
        if "fp32" in line:
            summ.update(fp32_ms_per_step=line["fp32"]["ms_per_step"], fp32_block_attn_ms=line["fp32"]["roofline"]["kernel_ms"],
                        fp32_mfma_busy_frac=line["fp32"]["roofline"]["mfma_busy_frac"])
        if "step_roofline" in line:
            summ["step_roofline_frac"] = line["step_roofline"]["frac"]
        if "cpu_baseline" in line:
            summ["cpu_points_per_s"] = line["cpu_baseline"]["value"]
        ordered["summary"] = summ
        print(json.dumps(ordered), flush=True)
    if multi:
        with c_stdout_to_stderr():
            dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
