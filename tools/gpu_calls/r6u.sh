#!/bin/bash
# round 6, call u: the suites that read through the handles, with the small path's executor in segments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_seeks.py tests/test_gpu_decoder_api.py tests/test_gpu_host_pipeline.py tests/test_gpu_levelc.py tests/test_gpu_exec_seg.py tests/test_gpu_seek_soak.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
