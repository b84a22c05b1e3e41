#!/bin/bash
# round 6, call aw: the phase clocks after calls aj ... av
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ZEEKSTD_AMD_LIB=$PWD/zeekstd_amd/libzk_clk.so python tools/exec_clocks.py 2048 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r6aw_exec_clocks.txt
cat gpurun_out/r6aw_exec_clocks.txt
