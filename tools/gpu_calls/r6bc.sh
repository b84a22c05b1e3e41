#!/bin/bash
# round 6, call bc: the Huffman decoder's four symbols per window as two pairs out of 32 bits (one 64-bit shift per window instead of eight): zk_k_huf before | after
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in wide "" wide ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"
done > gpurun_out/r6bc_huf_probe.txt 2>&1
cat gpurun_out/r6bc_huf_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py -q -x --timeout 900 2>&1 | tail -3
