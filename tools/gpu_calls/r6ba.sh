#!/bin/bash
# round 6, call ba: the own-table pass of a batch whose blocks share their frames' tables BEHIND the literal kernel instead of beside it: the step, before | after
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --cache /tmp/zkcache"
python bench.py $Q --steps 1 --warmup 0 > /dev/null 2>&1
for v in prev "" prev ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  ZEEKSTD_AMD_LIB=$PWD/$lib python bench.py $Q --sync 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v sync', d['value'], d['ms_per_step'])"
done > gpurun_out/r6ba_mopup_probe.txt 2>&1
cat gpurun_out/r6ba_mopup_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py -q -x --timeout 900 2>&1 | tail -3
