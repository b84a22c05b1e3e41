import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "zk_k" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "")[:22], r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0]
# last ~40 kernels
for s, e, n, q in rows[-44:]:
    print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:7.3f}  q{q:>3} {n}")
