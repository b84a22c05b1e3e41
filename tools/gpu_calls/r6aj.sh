#!/bin/bash
# round 6, call aj: four bytes per L1 lookup where a lane's group of four lies inside one run (one unaligned 4-byte load; the other lanes three more
# byte loads): byte gathers (-DZK_EXEC_WIDE=0) | this tree; then the parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in narrow "" narrow ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"
done > gpurun_out/r6aj_exec_probe.txt 2>&1
cat gpurun_out/r6aj_exec_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_exec_seg.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py -q -x --timeout 900 2>&1 | tail -4
