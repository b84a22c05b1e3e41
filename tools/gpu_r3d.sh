#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_decode.py tests/test_gpu_encoder_api.py -q -x --timeout 600 2>&1 | tail -5 > gpurun_out/r3_${tag}_tests.log; cat gpurun_out/r3_${tag}_tests.log
timeout 300 python tools/enc_probe.py 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_${tag}_enc_probe.log
timeout 300 python bench.py --no-c1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_${tag}_bench.log
