#!/bin/bash
# round 4, call b: how often the round-3 kernels fail (same seeks as call a), then the whole -m gpu suite with the new tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
ZEEKSTD_AMD_LIB=$PWD/tools/variants/libzk_r3.so timeout 300 python tools/seek_soak.py 128 30000 1 1 > gpurun_out/r4b_soak_r3.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r4b_tests.log 2>&1
tail -4 gpurun_out/r4b_soak_r3.log; tail -40 gpurun_out/r4b_tests.log
