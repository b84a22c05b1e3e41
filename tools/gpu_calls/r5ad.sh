#!/bin/bash
# round 5, call ad: after the status write-back fix -- kernels' verdicts against the simulator's, the shim against libzstd on damaged streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/scratch_gpu/verdicts.py 1500 3 2>&1 | tail -12
timeout 600 python tools/scratch_gpu/verdicts.py 1500 4 2>&1 | tail -12
timeout 900 python tools/fuzz_levelc_gpu.py 600 3 2>&1 | tail -40
