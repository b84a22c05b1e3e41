#!/bin/bash
# round 6, call cg: the full-size level-3 test; the encode fuzzer, 3000 cases with another seed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q --timeout 900 -k "default_level" 2>&1 | tail -3
timeout 1500 python tools/fuzz_encode_gpu.py 3000 777 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -5 | tee gpurun_out/r6cg_fuzz_encode.txt
