#!/bin/bash
# round 5, call h: the whole -m gpu suite on the tree with catch-up at every level, the cheap-offset rule, Level C, the gather ranks; then the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15
python tools/enc_probe.py 2048 2>&1 | tail -1
python bench.py > gpurun_out/r5h_bench.json 2> gpurun_out/r5h_bench.err; tail -c 1500 gpurun_out/r5h_bench.json; tail -3 gpurun_out/r5h_bench.err
