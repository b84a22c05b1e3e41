#!/bin/bash
# round 4, call e: the quad walk fed through LDS (feeder wave, ZkRevL): parity subset + the bench legs it changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_seeks.py tests/test_gpu_kernel_choice.py tests/test_gpu_decoder_api.py -m gpu -x -q > gpurun_out/r4e_tests.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-e2e --no-c1 --no-cpu-baseline > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
tail -5 gpurun_out/r4e_tests.log; tail -3 gpurun_out/r4e_bench.err
