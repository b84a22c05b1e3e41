#!/bin/bash
# round 6, call d: the fill pass with the segment's holes in LDS; segment sizes; the tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_exec_seg.py -x -q --timeout 600 2>&1 | tail -5
timeout 300 python tools/seg_probe.py --frames 1,5,16,64 --seg-kib 128,64,32 2>&1 | grep -v "verify 1" | tail -40 | tee gpurun_out/r6d_seg_probe_gpu.txt
rocprofv3 --kernel-trace -d gpurun_out/prof_r6d -- python tools/seg_probe.py --frames 1 --seg-kib 128,32 > /dev/null 2>&1
python tools/prof_timeline.py gpurun_out/prof_r6d 400 2>&1 | grep "seg_prep\|exec_seg\|exec_fill" | sort | uniq -c | sort -rn | head -30 > gpurun_out/r6d_timeline.txt; cat gpurun_out/r6d_timeline.txt
rm -rf gpurun_out/prof_r6d
