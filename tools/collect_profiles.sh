#!/bin/bash
# rocprofv3 evidence for profiles/ (run through gpurun; writes under gpurun_out/, summaries are copied into profiles/ by hand)
#   tools/collect_profiles.sh r04 [ref]      ref: also the reference-made archive legs (level 1 and 3, --archive libzstd)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() {   # $1 = suffix, rest = extra bench flags
  sfx=$1; shift
  # --sync --choice xxh64=5: one batch at a time with the checksums BEHIND the executor, so that every kernel runs alone and its average
  # duration in the trace is the one bench.py's HIP events measure (a lone batch would otherwise get zk_k_xxh64_follow beside the executor)
  common="--no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --choice xxh64=5 --no-fork --cache /tmp/zkcache $*"
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag$sfx -- python bench.py --steps 3 --warmup 1 $common > gpurun_out/prof_$tag$sfx.json 2> gpurun_out/prof_$tag$sfx.err
  python tools/prof_summary.py gpurun_out/prof_$tag$sfx 20 > gpurun_out/${tag}${sfx}_bench_c3_kernel_trace_stats.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_${tag}${sfx}_$c -o p --output-format csv -- python bench.py --steps 1 --warmup 1 $common > gpurun_out/pmc_${tag}${sfx}_$c.json 2> gpurun_out/pmc_${tag}${sfx}_$c.err
  done
  if [ -z "$sfx" ]; then python tools/pmc_summary.py gpurun_out/pmc_${tag}${sfx}_FETCH_SIZE gpurun_out/pmc_${tag}${sfx}_WRITE_SIZE c3 gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}${sfx}_pmc_fetch_write.txt
  else sec=reference_made_level_1; [ "$sfx" = "_ref3" ] && sec=reference_made_level_3
       python tools/pmc_summary.py gpurun_out/pmc_${tag}${sfx}_FETCH_SIZE gpurun_out/pmc_${tag}${sfx}_WRITE_SIZE c3 gpurun_out/${tag}_pmc_traffic.json $sec > gpurun_out/${tag}${sfx}_pmc_fetch_write.txt; fi
  echo "== $tag$sfx"; head -24 gpurun_out/${tag}${sfx}_bench_c3_kernel_trace_stats.txt; cat gpurun_out/${tag}${sfx}_pmc_fetch_write.txt
  rm -rf gpurun_out/prof_$tag$sfx gpurun_out/pmc_${tag}${sfx}_FETCH_SIZE gpurun_out/pmc_${tag}${sfx}_WRITE_SIZE     # raw traces: tens of MiB
}
one ""
if [ "$2" = "ref" ]; then
  one _ref1 --archive libzstd --level 1
  one _ref3 --archive libzstd --level 3
fi
