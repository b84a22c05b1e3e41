#!/bin/bash
# round 5, call aa: the old match kernel (levels >= 2) with the written-out walk, the neighbour-slot extension, the wave-uniform window skips
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_encoder_api.py -q --timeout 600 2>&1 | grep -E "passed|failed" | tail -2
python tools/enc_level_probe.py 1024 1 3 6 2>&1 | grep LEVEL
