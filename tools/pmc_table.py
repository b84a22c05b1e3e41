#!/usr/bin/env python3
"""Print the last dispatch's counter values per kernel from rocprofv3 --pmc csv directories."""
import csv, glob, os, sys, collections
tab = collections.OrderedDict()
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = {}
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:20]
            if "zk_k" not in k: continue
            key = (k, r["Counter_Name"]); did = int(r["Dispatch_Id"])
            agg.setdefault(key, {}); agg[key][did] = agg[key].get(did, 0.0) + float(r["Counter_Value"])
        for (k, c), v in agg.items():
            tab.setdefault(k, collections.OrderedDict())[c] = v[max(v)]
want = [a for a in os.environ.get("KERNELS", "").split(",") if a]
for k, cs in tab.items():
    if want and not any(w in k for w in want): continue
    print(k)
    for c, v in cs.items(): print(f"    {c:40s} {v:18.0f}")
