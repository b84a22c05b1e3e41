#!/bin/bash
# round 6, call a: the executor's tile width on a level-3 archive of the reference's (4 GiB): does a smaller resident set of frames pay?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 python tools/l3_exec_probe.py 4096 3 1 2>&1 | tail -20 | tee gpurun_out/r6a_l3_exec_probe.txt
