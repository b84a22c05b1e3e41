#!/bin/bash
# round 5, call an: the whole suite on the round's last tree (the bench line of that tree's kernels: profiles/r05e_bench.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|error" | tail -5 | tee gpurun_out/r05f_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee -a gpurun_out/r05f_gpu_tests.log
