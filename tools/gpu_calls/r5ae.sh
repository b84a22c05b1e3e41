#!/bin/bash
# round 5, call ae: the batch kernels' verdicts on damaged frames (checksums off) against the oracle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python tools/scratch_gpu/verdicts_batch.py 2>&1 | tail -40
