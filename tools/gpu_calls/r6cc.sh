#!/bin/bash
# round 6, call cc: phase clocks of the match kernel at level 2 (no dense candidates) and level 3 (DENSE instance)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lv in 2 3; do ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 1024 $lv 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids"; done | tee gpurun_out/r6cc_enc_clocks.txt
