#!/bin/bash
# round 6, call cf: the GPU encoder against its twin on 400 random structured inputs (levels 1 / 2 / 3 / 6 / 0 / 9, odd frame sizes, prefixes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_encode_gpu.py 400 20261001 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -5 | tee gpurun_out/r6cf_fuzz_encode.txt
