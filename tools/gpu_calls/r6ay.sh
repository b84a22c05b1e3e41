#!/bin/bash
# round 6, call ay: zk_k_fse_predef_fed with two walker waves of 32 lanes each (six waves per SIMD): before | after, then the decode tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in wide "" wide ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"
done > gpurun_out/r6ay_fse_half_probe.txt 2>&1
cat gpurun_out/r6ay_fse_half_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py tests/test_gpu_encode.py -q -x --timeout 900 2>&1 | tail -3
