#!/bin/bash
# round 5, call n: the stitch between the two barriers (behind the insertions whose LDS atomics nobody waited usefully for)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q --timeout 600 2>&1 | tail -2
python tools/enc_probe.py 2048 2>&1 | tail -1
python tools/enc_probe.py 2048 2>&1 | tail -1
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -19
