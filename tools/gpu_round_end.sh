#!/bin/bash
# what the driver runs at round end (full -m gpu suite, smoke, default bench) + the rocprofv3 evidence for profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03}
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12 > gpurun_out/${tag}_gpu_tests.log; cat gpurun_out/${tag}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/${tag}_bench.err
bash tools/collect_profiles.sh $tag
