#!/bin/bash
# round 5, call ab: damaged archives through the Level-C shim against the box's libzstd (same call sequence)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/fuzz_levelc_gpu.py 400 3 2>&1 | tail -8
