#!/bin/bash
# round 5, call b: phase clocks of zk_k_enc_match2 with the parse split into its sweeps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -19
