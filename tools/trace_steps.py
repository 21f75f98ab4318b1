"""Every forward of a rocprofv3 kernel trace, one line per step: start-to-start time, sum of kernel durations, sum of gaps,
and the per-kernel durations.  python tools/trace_steps.py <dir with *_kernel_trace.csv> [first-kernel-substring]"""
import csv
import glob
import sys

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prep_hash"
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(rows)])):
    t0 = int(rows[a]["Start_Timestamp"])
    nxt = int(rows[b]["Start_Timestamp"]) if b < len(rows) else int(rows[b - 1]["End_Timestamp"])
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[a:b]]
    print(f"step {n:3d}: {(nxt - t0) / 1e3:8.1f} us to next, kernels {sum(durs):7.1f} us, idle {(nxt - t0) / 1e3 - sum(durs):8.1f} | "
          + " ".join(f"{d:.1f}" for d in durs))
