#!/bin/bash
# round 6, call x: seeks into 64 KiB frames with a segment per block and the frame as ONE turn of the fill pass; the suites around it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_exec_seg.py tests/test_gpu_seeks.py tests/test_gpu_decoder_api.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -3
timeout 600 python tools/seek_probe.py 256 2000 2>&1 | tail -2 | tee gpurun_out/r6x_seek_probe.txt
