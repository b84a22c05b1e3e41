#!/bin/bash
# round 4, call g: walker trims (static ring slots in an unrolled round, baseline-only tables): parity subset, then the
# reference-made legs for the unrolled (product) and the rolled (tools/variants) walk
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_seeks.py tests/test_gpu_kernel_choice.py tests/test_gpu_decoder_api.py -m gpu -x -q > gpurun_out/r4g_tests.log 2>&1
Q="--steps 10 --warmup 3 --no-e2e --no-c1 --no-cpu-baseline"
timeout 300 python bench.py $Q > gpurun_out/r4g_bench_unrolled.json 2> gpurun_out/r4g_bench_unrolled.err
ZEEKSTD_AMD_LIB=$PWD/tools/variants/libzk_rolled.so timeout 300 python bench.py $Q > gpurun_out/r4g_bench_rolled.json 2> gpurun_out/r4g_bench_rolled.err
tail -3 gpurun_out/r4g_tests.log
for v in unrolled rolled; do python - <<PY
import json
j=json.loads(open('gpurun_out/r4g_bench_$v.json').read().strip().splitlines()[-1])
r=j['reference_made_archive']; s=j['seek']
print('$v', j['value'], 'ref', r['value'], r['kernel_ms']['zk_k_fse'], 'L3', r['level_3']['value'], r['level_3']['kernel_ms']['zk_k_fse'], 'seek', s['gpu_made_archive']['gpu_decoder_us']['p50'], s['reference_made_archive']['gpu_decoder_us']['p50'])
PY
done
