#!/bin/bash
# round 5, call af: the encoder against its twin over random shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/scratch_gpu/enc_fuzz.py 3000 2 2>&1 | tail -30
