#!/bin/bash
# round 5, call z: phase clocks of the old match kernel (levels >= 2: every position, lazy parse, in-frame far table) at level 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 1024 3 2>&1 | tail -19
