#!/bin/bash
# round 6, call k: bench.py at 2 / 8 / 3 ranks on the one GPU (shared-memory transport); ADVICE r5 fixes under their tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1500 python -m pytest tests/test_gpu_multirank_bench.py tests/test_gpu_gather_ranks.py -x -q --timeout 900 2>&1 | tail -3; done
timeout 1500 python -m pytest tests/test_gpu_levelc.py tests/test_gpu_decode.py tests/test_gpu_encode.py -x -q --timeout 900 2>&1 | tail -5
