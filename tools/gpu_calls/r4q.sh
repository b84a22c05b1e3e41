#!/bin/bash
# round 4, call q: the small-batch sequence walk with 8-byte cells -- seek tests, a quarter of the soak, the seek leg of the bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seeks.py tests/test_gpu_decode.py tests/test_gpu_decoder_api.py tests/test_gpu_kernel_choice.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 python -m pytest tests/test_gpu_seek_soak.py -x -q --timeout 600 -k "3-False or engine" 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-c1 --no-ref-level3 > gpurun_out/r04q_bench.json 2> gpurun_out/r04q_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04q_bench.json").read().strip().splitlines()[-1])
s=d["seek"]
for k in ("gpu_made_archive","reference_made_archive"):
    print(k, s[k]["gpu_decoder_us"], s[k].get("batches_of_1024"))
PY
