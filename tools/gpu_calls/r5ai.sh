#!/bin/bash
# round 5, call ai: the differential probes once more, long (rare races are what the short runs miss)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/scratch_gpu/verdicts.py 20000 11 2>&1 | tail -6
timeout 600 python tools/fuzz_levelc_gpu.py 4000 12 2>&1 | tail -6
timeout 600 python tools/scratch_gpu/enc_fuzz.py 8000 13 2>&1 | tail -6
timeout 600 python tools/scratch_gpu/verdicts2.py 3000 14 2>&1 | tail -8
