#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
timeout 1700 python -m pytest tests/test_cli.py tests/test_gpu_host_pipeline.py tests/test_gpu_encoder_api.py -m gpu -q --timeout 600 --durations=8 2>&1 | tail -40 > gpurun_out/r3_${tag}_tests.log; cat gpurun_out/r3_${tag}_tests.log
