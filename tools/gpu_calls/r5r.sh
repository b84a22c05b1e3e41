#!/bin/bash
# round 5, call r: the GPU encoder against its twin on 300 random structured inputs (sizes, levels, frame sizes, prefixes); the decode fuzz beside it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/fuzz_encode_gpu.py 300 5 2>&1 | tail -5
timeout 600 python tools/fuzz_decode_gpu.py 2>&1 | tail -3
