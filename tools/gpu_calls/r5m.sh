#!/bin/bash
# round 5, call m: hashes computed for every lane (no branch around them), one limit compare per position
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q --timeout 600 2>&1 | tail -2
python tools/enc_probe.py 2048 2>&1 | tail -1
python tools/enc_probe.py 2048 2>&1 | tail -1
