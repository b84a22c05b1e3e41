#!/bin/bash
# round 6, call q: the jump tables of both passes built together
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q -x --timeout 600 2>&1 | tail -2
python tools/enc_probe.py 2048 2>&1 | tail -1
python tools/enc_probe.py 2048 2>&1 | tail -1
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -19 | grep -v " - " | tee gpurun_out/r6q_enc_clocks.txt
