#!/bin/bash
# round 5, call s: single seeks with the executor pinned to 128 / 256 / 512 / 1024 lanes per frame
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/scratch_gpu/seek_lanes.py 2>&1 | grep exec_lanes
