#!/bin/bash
# round 4, call c: round-3 kernels' failure count on the soak; the two-wave quad walk: parity subset + the bench legs it changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
ZEEKSTD_AMD_LIB=$PWD/tools/variants/libzk_r3.so timeout 300 python tools/seek_soak.py 128 30000 1 1 > gpurun_out/r4c_soak_r3.log 2>&1
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_seeks.py tests/test_gpu_kernel_choice.py tests/test_gpu_full_size.py tests/test_gpu_decoder_api.py -m gpu -x -q > gpurun_out/r4c_tests.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-e2e --no-c1 --no-cpu-baseline > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err
tail -12 gpurun_out/r4c_soak_r3.log; tail -5 gpurun_out/r4c_tests.log; tail -3 gpurun_out/r4c_bench.err
