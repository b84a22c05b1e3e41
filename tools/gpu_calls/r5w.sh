#!/bin/bash
# round 5, call w: a seek's entropy stage as two kernels in a trace -- literals or sequences: which chain is the longer one?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_seek2 -- python tools/scratch_gpu/seek_split.py > gpurun_out/r5w.log 2>&1
grep split gpurun_out/r5w.log
python tools/prof_summary.py gpurun_out/prof_seek2 10 | cut -c1-120 | head -24
rm -rf gpurun_out/prof_seek2
