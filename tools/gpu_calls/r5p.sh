#!/bin/bash
# round 5, call p: where a single seek's 360 us go today: kernel trace of the sparse seek stream (64 MiB archive, 600 seeks)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_seek -- python tools/seek_probe.py 64 600 > gpurun_out/r5p_seek.log 2>&1
tail -3 gpurun_out/r5p_seek.log | cut -c1-400
python tools/prof_summary.py gpurun_out/prof_seek 24 | cut -c1-120
rm -rf gpurun_out/prof_seek
