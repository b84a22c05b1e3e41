#!/bin/bash
# round 6, call ch: phase clocks incl. the dense candidate kernel's three parts (level 3, level 9)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lv in 3 9; do ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 1024 $lv 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids\| 0.0 %"; done | tee gpurun_out/r6ch_enc_clocks.txt
