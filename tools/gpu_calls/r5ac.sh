#!/bin/bash
# round 5, call ac: the kernels' verdicts on damaged goldens against the simulator's (no checksums)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/scratch_gpu/verdicts.py 400 3 2>&1 | tail -40
