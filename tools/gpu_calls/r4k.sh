#!/bin/bash
# round 4, call k: in-frame far history of the encoder (level >= 2): encode / encoder-API / CLI / host-pipeline tests, encode speed per level
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_encoder_api.py tests/test_cli.py tests/test_gpu_host_pipeline.py -m gpu -x -q -s 2>&1 | tail -25 > gpurun_out/r4k_tests.log; cat gpurun_out/r4k_tests.log
timeout 300 python tools/level_probe.py > gpurun_out/r4k_levels.log 2>&1; cat gpurun_out/r4k_levels.log | grep level
