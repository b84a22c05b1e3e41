#!/bin/bash
# round 5, call a: the even-position matcher (zk_enc_match2.h) on hardware -- twin parity, timing, phase clocks; unaligned LDS reads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/ubench/ldsrd
timeout 900 python -m pytest tests/test_gpu_encode.py -q --timeout 600 2>&1 | tail -15
python tools/enc_probe.py 2048 2>&1 | tail -2
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 2048 1 2>&1 | tail -14
python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; tail -c 3000 gpurun_out/r5a_bench.json
