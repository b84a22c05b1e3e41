#!/bin/bash
# round 4, last call: the whole -m gpu suite + smoke + the driver's bench on the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | tail -20 > gpurun_out/r04e_gpu_tests.log; cat gpurun_out/r04e_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r04e_bench.err
