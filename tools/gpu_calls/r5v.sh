#!/bin/bash
# round 5, call v: the round's last tree -- what the driver runs at round end (suite, smoke, bench line) + the kernel trace and the FETCH / WRITE passes of the GPU-made leg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_round_end.sh r05b 2>&1 | tail -60
