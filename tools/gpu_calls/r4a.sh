#!/bin/bash
# round 4, call a: the small-batch defect -- round-3 kernels against the fixed companion loop, same seeks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
ZEEKSTD_AMD_LIB=$PWD/tools/variants/libzk_r3.so timeout 400 python tools/seek_soak.py 128 40000 1 1 > gpurun_out/r4a_soak_r3.log 2>&1
timeout 500 python tools/seek_soak.py 128 60000 1 1 > gpurun_out/r4a_soak_new.log 2>&1
timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_seeks.py -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
tail -3 gpurun_out/r4a_soak_r3.log gpurun_out/r4a_soak_new.log gpurun_out/r4a_tests.log
