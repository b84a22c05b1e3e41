#!/bin/bash
# round 6, call y: the kernels of one seek into a GPU-made 64 KiB frame (rocprofv3 kernel trace of the sparse seek stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/prof_r6y -- python tools/seek_probe.py 64 300 > /dev/null 2>&1
python - <<'PY' | tee gpurun_out/r6y_seek_kernels.txt
import glob, sqlite3, collections
for dbp in glob.glob("gpurun_out/prof_r6y/**/*_results.db", recursive=True):
    cur = sqlite3.connect(dbp).cursor()
    rows = sorted((r[1], r[2], r[0].split("(")[0].replace("void ", "")) for r in cur.execute("select name,start,end from kernels"))
    # the last 100 seeks: groups that start with zk_k_small_walk
    idx = [i for i, r in enumerate(rows) if r[2].startswith("zk_k_small_walk")]
    for tag, sel in (("with zk_k_exec_seg", True), ("frame executor", False)):
        dur = collections.defaultdict(list); spans = []
        for a, b in zip(idx[-220:-1], idx[-219:]):
            g = rows[a:b]
            if any("exec_seg" in x[2] for x in g) != sel: continue
            for s, e, n in g: dur[n].append((e - s) / 1e3)
            spans.append((g[-1][1] - g[0][0]) / 1e3)
        if not spans: continue
        print(f"-- {tag}: {len(spans)} seeks, first kernel's start to last kernel's end: median {sorted(spans)[len(spans)//2]:.1f} us")
        for n, v in dur.items(): print(f"   {n[:40]:40s} median {sorted(v)[len(v)//2]:7.1f} us  x{len(v) / len(spans):.1f}")
PY
rm -rf gpurun_out/prof_r6y
