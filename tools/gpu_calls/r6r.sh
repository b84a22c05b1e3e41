#!/bin/bash
# round 6, call r: the executor touching far sources a tile or two ahead (FAR), level-3 and level-1 archives of the reference's
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 python tools/l3_exec_probe.py 4096 3 1 2>&1 | tail -25 | tee gpurun_out/r6r_l3_far_probe.txt
