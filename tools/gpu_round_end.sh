#!/bin/bash
# what the driver runs at round end (full -m gpu suite, smoke, default bench) + the rocprofv3 evidence for profiles/
#   tools/gpu_round_end.sh r04 [ref]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r04}
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=8 2>&1 | tail -24 > gpurun_out/${tag}_gpu_tests.log; cat gpurun_out/${tag}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/${tag}_bench.err
# the caches of the profiling passes (forked generation, no profiler attached)
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --cache /tmp/zkcache"
python bench.py $Q > /dev/null 2>&1
if [ "$2" = "ref" ]; then python bench.py $Q --archive libzstd --level 1 > /dev/null 2>&1; python bench.py $Q --archive libzstd --level 3 > /dev/null 2>&1; fi
bash tools/collect_profiles.sh $tag $2
