#!/bin/bash
# round 6, call ap: the whole GPU suite on the tree with the executor of calls ac ... al and the fed checksum pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -6 > gpurun_out/r6ap_gpu_tests.log
cat gpurun_out/r6ap_gpu_tests.log
