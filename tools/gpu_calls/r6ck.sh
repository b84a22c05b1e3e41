#!/bin/bash
# round 6, call ck: what the driver runs at round end on the last tree (the -m gpu suite, smoke, the default bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=5 2>&1 | tail -12 > gpurun_out/r06f_gpu_tests.log; cat gpurun_out/r06f_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err; echo "bench rc=$?"; tail -4 gpurun_out/r06f_bench.err
