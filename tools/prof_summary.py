#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db: per-kernel totals + the last N dispatches."""
import sqlite3, sys, glob
path = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
for dbp in dbs:
    cur = sqlite3.connect(dbp).cursor()
    print("==", dbp)
    print(f"{'kernel':40s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name.split('(')[0][:40]:40s} {calls:6d} {tot/1e3:12.1f} {avg/1e3:10.2f} {pct:6.2f}")
    rows = list(cur.execute("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by start"))
    print(f"-- last {last} dispatches: name dur_us grid wg vgpr sgpr lds scratch")
    for r in rows[-last:]:
        print(f"{r[0].split('(')[0][:32]:32s} {r[1]/1e3:10.1f} {r[2]:8d} {r[3]:5d} {r[4]:4d} {r[5]:4d} {r[6]:7d} {r[7]:5d}")
