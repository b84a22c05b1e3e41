#!/bin/bash
# round 5, call k: the match kernels per level (2 GiB): what catch-up + the cheap-offset rule cost the old kernel at levels 3 / 6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/enc_level_probe.py 1024 1 3 6 2>&1 | grep LEVEL
