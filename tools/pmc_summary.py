"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch, per kernel and counter."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
kernels = sorted({k for k, _ in acc})
for k in kernels:
    print(k)
    for (kk, c), (s, n) in sorted(acc.items()):
        if kk == k:
            print(f"    {c:32s} {s/n:16.1f}   (n={n})")
