#!/bin/bash
# round 6, call au: the executor's tile width again, on the kernel as it is now: 128 / 256 / 512 / 1024 lanes, 2048 frames
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in 128 256 512 1024; do
  ZK_LANES=$l python tools/exec_probe.py 2048 2>&1 | grep EXECVAR
done > gpurun_out/r6au_lanes_probe.txt 2>&1
cat gpurun_out/r6au_lanes_probe.txt
