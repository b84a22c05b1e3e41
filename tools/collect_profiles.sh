#!/bin/bash
# rocprofv3 evidence for profiles/ (run through gpurun; writes under gpurun_out/, summaries are copied into profiles/ by hand)
#   tools/collect_profiles.sh r03
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 3 --warmup 1 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --no-fork"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py $common > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
python tools/prof_summary.py gpurun_out/prof_$tag 20 > gpurun_out/${tag}_bench_c3_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_${tag}_$c -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --no-fork > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.err
done
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE c3 gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_fetch_write.txt
cat gpurun_out/${tag}_bench_c3_kernel_trace_stats.txt | head -34
cat gpurun_out/${tag}_pmc_fetch_write.txt
