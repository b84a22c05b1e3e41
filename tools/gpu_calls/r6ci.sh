#!/bin/bash
# round 6, call ci: a prefix too short to leave history (the far tables stay out); the encode tests; 1500 fuzz cases incl. 1- and 3-byte prefixes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encode.py -x -q --timeout 900 2>&1 | tail -3
timeout 1500 python tools/fuzz_encode_gpu.py 1500 31 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -5 | tee gpurun_out/r6ci_fuzz_encode.txt
