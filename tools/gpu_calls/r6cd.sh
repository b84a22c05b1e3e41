#!/bin/bash
# round 6, call cd: level 3 -- the encode tests, phase clocks of the match kernel, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encode.py -x -q --timeout 900 2>&1 | tail -2
ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py 1024 3 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids\| 0.0 %" | tee gpurun_out/r6cd_enc_clocks.txt
bash tools/gpu_calls/r6cb.sh 2>&1 | grep "zk_k_enc_dense\|zk_k_enc_match"; grep ratio gpurun_out/r6cb_l3.txt
