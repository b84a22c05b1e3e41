#!/usr/bin/env python3
"""Start / end of every kernel of the last N decodes in a rocprofv3 --kernel-trace results .db, relative to each decode's first kernel
(zk_k_walk): which kernels overlap, what the step waits for.   prof_timeline.py <dir or .db> [decodes]"""
import glob
import sqlite3
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
for dbp in dbs:
    cur = sqlite3.connect(dbp).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    end = "end" if "end" in cols else None
    q = f"select name,start,{end if end else 'start+duration'} from kernels order by start"
    rows = [(r[0].split("(")[0].replace("void ", ""), r[1], r[2]) for r in cur.execute(q) if "zk_k_" in r[0]]
    # a decode opens with the counting walk: zk_k_walk, zk_k_scan, zk_k_walk
    opens = [i for i, r in enumerate(rows) if r[0] == "zk_k_walk" and i + 1 < len(rows) and rows[i + 1][0] == "zk_k_scan"]
    for k, i in enumerate(opens[-n:]):
        j = opens[opens.index(i) + 1] if opens.index(i) + 1 < len(opens) else len(rows)
        t0 = rows[i][1]
        print(f"-- decode {k} ({dbp.split('/')[-1]})")
        for name, s, e in rows[i:j]:
            print(f"   {name[:34]:34s} {(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})")
