#!/bin/bash
# round 5, call aj: behind the Block_Maximum_Size check -- the new tests, then the long probes with other seeds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_levelc.py tests/test_gpu_encode.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5
timeout 600 python tools/scratch_gpu/verdicts2.py 6000 21 2>&1 | tail -8
timeout 600 python tools/fuzz_levelc_gpu.py 6000 22 2>&1 | tail -8
timeout 600 python tools/scratch_gpu/verdicts.py 20000 23 2>&1 | tail -4
