"""Print a rocprofv3 kernel_stats.csv compactly."""
import csv, sys, glob
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in rows[:int(20)]:
            print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
