#!/bin/bash
# round 4, call f: does a 64-register checksum kernel overlap the other batch's executor?  headline step under xxh64 = wide (default) / lean
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernel_choice.py -m gpu -x -q > gpurun_out/r4f_tests.log 2>&1
Q="--steps 20 --warmup 3 --no-e2e --no-c1 --no-cpu-baseline --no-seek --no-ref-archive"
timeout 300 python bench.py $Q > gpurun_out/r4f_bench_wide.json 2> gpurun_out/r4f_bench_wide.err
timeout 300 python bench.py $Q --choice xxh64=3 > gpurun_out/r4f_bench_lean.json 2> gpurun_out/r4f_bench_lean.err
timeout 300 python bench.py $Q --choice xxh64=1 > gpurun_out/r4f_bench_narrow.json 2> gpurun_out/r4f_bench_narrow.err
tail -3 gpurun_out/r4f_tests.log
for v in wide lean narrow; do python - <<PY
import json
j=json.loads(open('gpurun_out/r4f_bench_$v.json').read().strip().splitlines()[-1])
print('$v', j['value'], j['ms_per_step'], 'one', j['one_batch_at_a_time']['ms_per_step'], j['roofline']['kernel_ms'])
PY
done
