#!/bin/bash
# round 6, call c: the segmented executor's tests + where its time goes (kernel trace of a lone 2 MiB frame and of 16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_exec_seg.py -x -q --timeout 600 2>&1 | tail -15
rocprofv3 --kernel-trace -d gpurun_out/prof_r6c -- python tools/seg_probe.py --frames 1,16 > gpurun_out/r6c_probe.txt 2>&1
python tools/prof_timeline.py gpurun_out/prof_r6c 40 2>&1 | grep -B1 -A12 "seg_prep" | tail -120 > gpurun_out/r6c_timeline.txt; tail -60 gpurun_out/r6c_timeline.txt
rm -rf gpurun_out/prof_r6c
