#!/bin/bash
# round 5, call ao: generated frames through the kernels, seeds the suite does not use
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 110 python tools/scratch_gpu/gen_campaign.py 24 3000000 2>&1 | tail -12
