#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
timeout 600 python -m pytest tests/test_gpu_encode.py -q -x --timeout 600 2>&1 | tail -3 > gpurun_out/r3_${tag}_enc_tests.log; cat gpurun_out/r3_${tag}_enc_tests.log
timeout 300 python tools/enc_probe.py 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_${tag}_enc_probe.log
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so timeout 300 python tools/enc_clocks.py 2048 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_${tag}_clocks.log
