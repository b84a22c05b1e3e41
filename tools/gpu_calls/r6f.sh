#!/bin/bash
# round 6, call f: the LDS fill with the whole segment as an image (bulk load / store), word reads + byte writes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_exec_seg.py -x -q --timeout 600 2>&1 | tail -5
timeout 300 python tools/seg_probe.py --frames 1,5,16,64 --seg-kib 128,64 2>&1 | grep -v "verify 1" | tail -40 | tee gpurun_out/r6f_seg_probe_gpu.txt
timeout 300 python tools/seg_probe.py --frames 1,16 --seg-kib 128 --archive libzstd --level 3 2>&1 | grep -v "verify 1" | tail -40 | tee gpurun_out/r6f_seg_probe_l3.txt
rocprofv3 --kernel-trace -d gpurun_out/prof_r6f -- python tools/seg_probe.py --frames 1 --seg-kib 128,64 > /dev/null 2>&1
python tools/prof_timeline.py gpurun_out/prof_r6f 400 2>&1 | grep "seg_prep\|exec_seg\|exec_fill" | awk '{print $1, $NF, $(NF-1)}' | sort | uniq -c | sort -rn | head -30 > gpurun_out/r6f_timeline.txt; cat gpurun_out/r6f_timeline.txt
rm -rf gpurun_out/prof_r6f
