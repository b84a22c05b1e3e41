#!/bin/bash
# round 4, call i: line-aligned stores (Huffman bursts on 32-byte boundaries, a block's records on a 64-byte line): parity subset,
# kernel times, WRITE_SIZE of the GPU-made leg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_seeks.py tests/test_gpu_kernel_choice.py tests/test_gpu_decoder_api.py tests/test_gpu_host_pipeline.py -m gpu -x -q > gpurun_out/r4i_tests.log 2>&1
tail -3 gpurun_out/r4i_tests.log
Q="--no-cpu-baseline --no-seek --no-e2e --no-c1 --cache /tmp/zkcache"
timeout 300 python bench.py --steps 10 --warmup 3 $Q > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
python - <<PY
import json
j=json.loads(open('gpurun_out/r4i_bench.json').read().strip().splitlines()[-1])
r=j['reference_made_archive']
print('value', j['value'], j['ms_per_step'], j['roofline']['kernel_ms']); print('ref', r['value'], r['kernel_ms']); print('L3', r['level_3']['value'], r['level_3']['kernel_ms'])
PY
P="--steps 1 --warmup 1 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --no-fork --cache /tmp/zkcache"
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --cache /tmp/zkcache > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_r4i_$c -o p --output-format csv -- python bench.py $P > gpurun_out/pmc_r4i_$c.json 2> gpurun_out/pmc_r4i_$c.err
done
python tools/pmc_summary.py gpurun_out/pmc_r4i_FETCH_SIZE gpurun_out/pmc_r4i_WRITE_SIZE c3 gpurun_out/r4i_pmc_traffic.json > gpurun_out/r4i_pmc_fetch_write.txt
cat gpurun_out/r4i_pmc_fetch_write.txt
rm -rf gpurun_out/pmc_r4i_FETCH_SIZE gpurun_out/pmc_r4i_WRITE_SIZE
