#!/bin/bash
# round 4, call p: configs[0] through the handles with the first read decoding a batch ahead; the tests of the Decoder
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder_api.py tests/test_gpu_seeks.py tests/test_gpu_host_pipeline.py tests/test_cli.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3
python - <<'PY'
import bench, zeekstd_amd as zk, numpy as np
eng = zk.Engine(0)
r = bench.small_input_leg(eng, zk)
print("configs0", r["decoder"], r["encoder"], r.get("cpu_1thread"))
PY
