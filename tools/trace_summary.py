"""Timeline of one steady-state forward from a rocprofv3 kernel trace: start offset, duration, gap to the previous
kernel's end, stream.  python tools/trace_summary.py <dir with *_kernel_trace.csv> [first-kernel-substring]"""
import csv
import glob
import sys

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prep_hash"
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 4:
    raise SystemExit(f"found {len(starts)} steps")
a, b = starts[-3], starts[-2]   # the second-to-last complete step
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print(f"step of {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next step's first kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q={q}  {name}")
    prev_end = max(prev_end, e)
print(f"  end of last kernel: +{(prev_end - t0) / 1e3:.1f} us")
