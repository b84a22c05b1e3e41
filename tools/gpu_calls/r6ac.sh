#!/bin/bash
# round 6, call ac: the executor's gathers as GLOBAL loads off two folded bases (5 instead of 12 vector instructions per byte), pointers made
# uniform through their offsets (no flat loads / stores): HEAD's library | the old gather with global pointers | the new gather
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base gold ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  for r in 1 2; do ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"; done
done > gpurun_out/r6ac_exec_probe.txt 2>&1
cat gpurun_out/r6ac_exec_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_exec_seg.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py -q -x --timeout 900 2>&1 | tail -4
