#!/bin/bash
# round 6, call ce: the match kernel's DENSE instance, phase clocks of three builds: as is / without the look-ahead rule (output differs: timing only) /
# the candidate entries requested behind the stitch instead of at the top of the group
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in clk clkna clkdl; do echo "== $v"; ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_$v.so python tools/enc_clocks.py 1024 3 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids\| 0.0 %"; done | tee gpurun_out/r6ce_enc_clocks.txt
