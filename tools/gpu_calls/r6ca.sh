#!/bin/bash
# round 6, call ca: dense far history (level 0 / >= 3): the encode tests against the twin, ratios and encode rates by level
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encode.py tests/test_gpu_encoder_api.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12
timeout 600 python tools/level_probe.py 2>&1 | tee gpurun_out/r6ca_level_probe.txt | tail -8
