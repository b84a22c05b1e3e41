#!/bin/bash
# round 6, call an: the checksum pass of 2048 verified frames: a wave per frame (1) | sixteen frames per wave (2) | 64 chains fed by two waves (4) | 16 chains fed by one wave (5 = the new default)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for x in 2 5 6 2 5 6; do
  ZK_XXH=$x python tools/exec_probe.py 2048 verify 2>&1 | grep EXECVAR
done > gpurun_out/r6an_xxh_probe.txt 2>&1
cat gpurun_out/r6an_xxh_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py tests/test_gpu_encode.py -q -x --timeout 900 2>&1 | tail -3
