import csv, glob, sys, collections
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 40 calls of the first workload: find coord_keys starts
starts=[i for i,r in enumerate(rows) if "coord_keys" in r["Kernel_Name"]]
a,b=starts[20],starts[21]
t0=int(rows[a]["Start_Timestamp"]); prev=t0; busy=0
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); busy+=e-s
    print(f"+{(s-t0)/1e3:8.1f} dur {(e-s)/1e3:6.1f} gap {(s-prev)/1e3:6.1f} {r['Kernel_Name'][:70]}")
    prev=e
print("call period", (int(rows[b]["Start_Timestamp"])-t0)/1e3, "busy", busy/1e3, "kernels", b-a)
