#!/bin/bash
# round 5, call al: the whole suite and the driver's bench line on the round's last commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|error" | tail -5 | tee gpurun_out/r05e_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 900 python bench.py > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err; echo "bench rc=$?"
