#!/bin/bash
# round 3, first GPU pass: encoder parity tests, encode kernel times, other data shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encode.py -q -x --timeout 600 > gpurun_out/r3a_enc_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_enc_tests.log
tail -5 gpurun_out/r3a_enc_tests.log
timeout 300 python tools/enc_probe.py 2048 > gpurun_out/r3a_enc_probe.log 2>&1; tail -3 gpurun_out/r3a_enc_probe.log
timeout 600 python tools/data_probe.py 1024 > gpurun_out/r3a_data_probe.log 2>&1; tail -12 gpurun_out/r3a_data_probe.log
