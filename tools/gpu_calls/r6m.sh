#!/bin/bash
# round 6, call m: bench.py's per-rank host setup at world 8 at full size (eight processes x 4 GiB of generator text; no GPU work)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
free -g | head -2
timeout 500 python tools/setup_probe.py 8 2048 2>&1 | tail -2 | tee gpurun_out/r6m_setup_world8.txt
