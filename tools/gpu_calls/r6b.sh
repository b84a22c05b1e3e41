#!/bin/bash
# round 6, call b: first run of the executor in segments -- parity on small batches of 2 MiB frames and its time against one workgroup per frame
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/seg_probe.py --frames 1,5,16,64 2>&1 | tail -40 | tee gpurun_out/r6b_seg_probe_gpu.txt
timeout 300 python tools/seg_probe.py --frames 1,5,16 --archive libzstd 2>&1 | tail -30 | tee gpurun_out/r6b_seg_probe_libzstd.txt
