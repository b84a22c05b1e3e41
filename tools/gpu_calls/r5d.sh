#!/bin/bash
# round 5, call d: aligned reads again; capped matches take the length of the slot 16 bytes on; the walk written out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q --timeout 600 2>&1 | tail -3
python tools/enc_probe.py 2048 2>&1 | tail -1
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -19
