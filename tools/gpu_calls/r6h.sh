#!/bin/bash
# round 6, call h: do the checksum waves follow the fill pass?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/seg_probe.py --frames 1,16 --seg-kib 128 2>&1 | grep "verify 1" | tee gpurun_out/r6h_seg_probe_gpu.txt
rocprofv3 --kernel-trace -d gpurun_out/prof_r6h -- python tools/seg_probe.py --frames 16 --seg-kib 128 > /dev/null 2>&1
python tools/prof_timeline.py gpurun_out/prof_r6h 400 2>&1 | grep -B8 -A6 "xxh64_follow" | head -150 > gpurun_out/r6h_timeline.txt; grep -c follow gpurun_out/r6h_timeline.txt; grep -A14 "exec_seg<1024" gpurun_out/r6h_timeline.txt | head -60
rm -rf gpurun_out/prof_r6h
