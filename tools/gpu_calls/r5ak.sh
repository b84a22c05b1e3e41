#!/bin/bash
# round 5, call ak: frames drawn from the whole format (tests/helpers/zstd_gen.py) through the kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_generated_frames.py -m gpu -q -x 2>&1 | tail -25
