#!/bin/bash
# round 5, call t: the root-side check of bench.py's gather leg, exercised on one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/scratch_gpu/gather_remote_check.py 2>&1 | tail -3
