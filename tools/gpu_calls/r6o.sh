#!/bin/bash
# round 6, call o: the matcher's walk eight steps per turn (jump tables by ds_bpermute, ZKE_WALK8): twin identity and its time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encode.py -q -x --timeout 600 2>&1 | tail -3
python tools/enc_probe.py 2048 2>&1 | tail -1
python tools/enc_probe.py 2048 2>&1 | tail -1
