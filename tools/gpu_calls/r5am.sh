#!/bin/bash
# round 5, call am: Level C behind ZSTD_d_windowLogMax
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_generated_frames.py -m gpu -q -x -k decoder 2>&1 | tail -12
