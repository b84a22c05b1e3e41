#!/bin/bash
# round 5, call f: the Level-C shim (libzstd's symbols over the engine), zk_frame_content_sizes, the crafted-overflow seek, decode suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_levelc.py -q --timeout 600 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_seeks.py tests/test_gpu_decode.py tests/test_gpu_decoder_api.py -q --timeout 600 2>&1 | tail -8
