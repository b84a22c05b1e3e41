#!/bin/bash
# round 6, call s: FAR with its touches landing in LDS: the kernel-choice tests, then the level-3 / level-1 probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernel_choice.py -x -q --timeout 600 -k "far or unknown" 2>&1 | tail -3
timeout 500 python tools/l3_exec_probe.py 4096 3 1 2>&1 | tail -16 | tee gpurun_out/r6s_l3_far_probe.txt
