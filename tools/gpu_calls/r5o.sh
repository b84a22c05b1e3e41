#!/bin/bash
# round 5, call o: the previous-offset window only where a lane of the wave has four equal bytes at it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q --timeout 600 2>&1 | tail -2
python tools/enc_probe.py 2048 2>&1 | tail -1
python tools/enc_probe.py 2048 2>&1 | tail -1
