#!/bin/bash
# round 6, call i: configs[0]'s Decoder leg, executor per frame / in segments; its kernels in a trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/c0_probe.py 2>&1 | tail -8 | tee gpurun_out/r6i_c0_probe.txt
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_r6i -- python tools/c0_probe.py 3 > /dev/null 2>&1
python - <<'PY' | tee gpurun_out/r6i_c0_trace.txt
import glob, sqlite3
for dbp in glob.glob("gpurun_out/prof_r6i/**/*_results.db", recursive=True):
    cur = sqlite3.connect(dbp).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    rows = [(r[1], r[2], r[0].split("(")[0].replace("void ", "")) for r in cur.execute("select name,start,end from kernels")]
    mc = [t for t in tabs if "memory_cop" in t.lower()]
    for t in mc[:1]:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        try:
            rows += [(r[0], r[1], "memcpy " + str(r[2])) for r in cur.execute(f"select start,end,name from {t}")]
        except Exception as ex:
            print("memcpy table", t, cols, ex)
    rows.sort()
    # the last decode of the "by shape" variant: find the last group of kernels that contains zk_k_exec_seg
    idx = [i for i, r in enumerate(rows) if "zk_k_seg_prep" in r[2]]
    if not idx: continue
    i = idx[len(idx) // 2]
    lo = max(0, i - 14); t0 = rows[lo][0]
    for s, e, nme in rows[lo:i + 16]:
        print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms ({(e - s) / 1e6:7.3f})  {nme[:60]}")
PY
rm -rf gpurun_out/prof_r6i
