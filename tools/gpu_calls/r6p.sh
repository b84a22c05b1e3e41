#!/bin/bash
# round 6, call p: phase clocks of zk_k_enc_match2 with the walk eight steps per turn
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -19 | tee gpurun_out/r6p_enc_clocks.txt
