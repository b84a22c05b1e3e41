#!/bin/bash
# round 5, call ag: verdicts on damaged prefix archives, zk_frame_content_sizes on damaged frames, the Decoder over damaged archives
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/scratch_gpu/verdicts2.py 600 5 2>&1 | tail -40
