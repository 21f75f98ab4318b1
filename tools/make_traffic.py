"""profiles/attn_traffic.json from the PMC passes of tools/pmc.sh (per-launch figures of the block-attention kernel).

traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-B read requests at 64 B, so it is doubled as MI355X_MICROARCH.md §HBM prescribes; WRITE_SIZE is exact).
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the matrix pipe's busy cycles summed over the
1024 SIMDs, over the kernel's duration in cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs) times 1024 SIMDs.
python tools/make_traffic.py <pmc dir bf16> <pmc dir fp32> <out.json>
"""
import collections
import csv
import glob
import hashlib
import json
import os
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

WANT = ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
out = {"source_sha256": source_sha256(), "git_head": git_head() or os.environ.get("HEPT_GIT_HEAD"),
       "command": "tools/pmc.sh (rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py --steps 10 "
                  "--warmup 3 --no-cpu-baseline --no-extra [--precision fp32])"}
for prec, root in (("bf16", sys.argv[1]), ("fp32", sys.argv[2])):
    acc = collections.defaultdict(lambda: [0.0, 0])
    names = set()
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            # the dominant kernel only: block_attn_kernel (16-bit tiles) / block_attn_split_kernel (f32 tiles)
            if ("block_attn_kernel" in name or "block_attn_split_kernel" in name) and r["Counter_Name"] in WANT:
                names.add(name)
                a = acc[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    m = {k: v[0] / v[1] for k, v in acc.items()}
    out[prec] = 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    busy = None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    out[prec + "_mfma_busy_frac"] = busy
    out[prec + "_kernel"] = sorted(names)[0] if len(names) == 1 else sorted(names)   # the template the counters belong to
    out[prec + "_detail"] = {"FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
                             "TCC_EA0_RDREQ": m.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ": m.get("TCC_EA0_WRREQ_sum"),
                             "SQ_VALU_MFMA_BUSY_CYCLES": m.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                             "GRBM_GUI_ACTIVE": m.get("GRBM_GUI_ACTIVE"),
                             "formula": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per launch; "
                                        "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
