#!/bin/bash
# round 6, call cb: kernel trace of one level-3 encode of 4 GiB (the two dense kernels and the match kernel apart)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sed 's/for lvl in (1, 2, 3, 6, 9):/for lvl in (3,):/' tools/level_probe.py > /tmp/l3_probe.py
cp /tmp/l3_probe.py tools/_l3_probe.py
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l3 -- python tools/_l3_probe.py > gpurun_out/r6cb_l3.txt 2>&1
python tools/prof_summary.py gpurun_out/prof_l3 6 > gpurun_out/r6cb_l3_kernel_trace_stats.txt; cat gpurun_out/r6cb_l3_kernel_trace_stats.txt | head -20
rm -rf gpurun_out/prof_l3 tools/_l3_probe.py
