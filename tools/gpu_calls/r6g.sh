#!/bin/bash
# round 6, call g: checksums beside the fill pass (its progress words); the policy by shape; the decode suites that touch small batches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_exec_seg.py tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py -x -q --timeout 600 2>&1 | tail -5
timeout 300 python tools/seg_probe.py --frames 1,5,16,32 --seg-kib 128 2>&1 | tail -40 | tee gpurun_out/r6g_seg_probe_gpu.txt
