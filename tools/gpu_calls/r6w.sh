#!/bin/bash
# round 6, call w: the whole -m gpu suite, smoke, the driver's bench line (mid-round check)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -24 > gpurun_out/r06a_gpu_tests.log; cat gpurun_out/r06a_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r06a_bench.err
cut -c1-900 gpurun_out/r06a_bench.json
