#!/bin/bash
# round 4, call r: the pipeline's chunks without the checksum waves beside their executors -- host-pipeline tests, the end-to-end leg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_decoder_api.py tests/test_gpu_kernel_choice.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3
python - <<'PY'
import bench, zeekstd_amd as zk, numpy as np
from oracle import zko
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(2048 * bench.FRAME, 0), np.uint8)
r = bench.end_to_end(eng, zk, data, 2048, True, reps=3)
print("e2e 4 GiB", r["decode"], r["encode"])
print("configs0", bench.small_input_leg(eng, zk)["decoder"])
PY
