"""profiles/attn_traffic.json from the PMC passes of tools/pmc.sh (per-launch HBM bytes of block_attn_kernel).

traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-B read requests at 64 B, so it is doubled as MI355X_MICROARCH.md §HBM prescribes; WRITE_SIZE is exact).
"""
import csv, glob, json, sys, collections
out = {}
for prec, root in (("bf16", sys.argv[1]), ("fp32", sys.argv[2])):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "block_attn_kernel" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    m = {k: v[0] / v[1] for k, v in acc.items()}
    out[prec] = 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    out[prec + "_detail"] = {"FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
                             "TCC_EA0_RDREQ": m.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ": m.get("TCC_EA0_WRREQ_sum"),
                             "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes per block_attn launch"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
