#!/bin/bash
# round 6, call ad: the chase without its last pass (a min over the lane's words says whether any still points into the tile) + run marks as
# DELTAS (one addition per byte in the slot pass, overlapping matches through a bitmap of slots): the library of call ac | this tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  for r in 1 2; do ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"; done
done > gpurun_out/r6ad_exec_probe.txt 2>&1
cat gpurun_out/r6ad_exec_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_exec_seg.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py -q -x --timeout 900 2>&1 | tail -4
