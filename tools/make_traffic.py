"""profiles/attn_traffic.json from PMC passes (tools/pmc_all.sh): per-launch figures of the block-attention kernel, one
record per (WORKLOAD, kernel template) -- round 5 keyed the records by template only, and the batched tracking-6k
clouds' pass overwrote the headline's (same template, other workload).

traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-B read requests at 64 B, so it is doubled as MI355X_MICROARCH.md section HBM prescribes; WRITE_SIZE is exact).
These are the L2's memory-side requests: hits in the 256 MB Infinity Cache are counted (same guide), so `traffic` is
what leaves the L2s, an upper bound of what reaches HBM.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the matrix pipe's busy cycles summed over the
1024 SIMDs, over the kernel's duration in cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs) times 1024 SIMDs.
valu_issue_frac = 4 * SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 * 1024): a wave64 vector instruction occupies its SIMD's
vector ALU for 4 cycles.
python tools/make_traffic.py <out.json> <key>=<pmc dir> [...]     key = bench.py's record key + "/" + precision, e.g. c3/bf16
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha256():
    """Digest of the sources the block-attention kernels are compiled from: bench.py refuses a record whose digest is
    not that of the tree it runs in (a stale PMC pass must not be passed off as this build's traffic)."""
    h = hashlib.sha256()
    for rel in ("hept_amd/csrc/block_attn.hip", "hept_amd/csrc/common.h", "hept_amd/csrc/p2p_dev.h", "hept_amd/csrc/Makefile"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


def template_key(name):
    """'void (anonymous namespace)::block_attn_kernel<4, true, ...>(char const*, ...)' -> 'block_attn_kernel<4,true,...>'"""
    m = re.search(r"(block_attn(?:_split)?_kernel<[^>]*>)", name)
    return m.group(1).replace(" ", "") if m else None


WANT = ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum",
        "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
        "SQ_WAIT_ANY")
out = {"source_sha256": source_sha256(), "git_head": git_head() or os.environ.get("HEPT_GIT_HEAD"),
       "formula": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch (L2 memory-side requests: Infinity-Cache hits "
                  "included); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); valu_issue_frac = "
                  "4 * SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 * 1024)",
       "by_workload": {}}
for spec in sys.argv[2:]:
    wkey, root = spec.split("=", 1)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    full = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = template_key(r["Kernel_Name"])
            if key and r["Counter_Name"] in WANT:
                full[key] = r["Kernel_Name"]
                a = acc[key][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for key, cnt in acc.items():
        m = {k: v[0] / v[1] for k, v in cnt.items()}
        if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        cycles = m["GRBM_GUI_ACTIVE"] / 8 if m.get("GRBM_GUI_ACTIVE") else None
        out["by_workload"].setdefault(wkey, {})[key] = {
            "kernel": full[key], "command": "rocprofv3 --pmc <one group per pass> --kernel-trace, passes under " + os.path.basename(root.rstrip("/")),
            "traffic": 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024,
            "write_bytes": m["WRITE_SIZE"] * 1024,
            "l2_hit_frac": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if m.get("TCC_MISS_sum") else None,
            "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024) if cycles and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None,
            "valu_issue_frac": 4 * m["SQ_INSTS_VALU"] / (cycles * 1024) if cycles and "SQ_INSTS_VALU" in m else None,
            "wait_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m else None,
            "kernel_cycles": cycles, "counters": m}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({w: {k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in t.items()}
                  for w, t in out["by_workload"].items()}, indent=1))
