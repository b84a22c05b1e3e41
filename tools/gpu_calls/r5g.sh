#!/bin/bash
# round 5, call g: zk_gather_seekable between 2 / 3 / 8 processes on one GPU (shared-memory collectives), then sharded decode of the one archive
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_gather_ranks.py -q --timeout 900 -x 2>&1 | tail -30
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py -q --timeout 600 -k "gather or shard" 2>&1 | tail -3
