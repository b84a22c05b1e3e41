#!/bin/bash
# round 6, call z: frame lists / device pointers / two batches in flight / short-frame seeks with the executor in segments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_exec_seg.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12
