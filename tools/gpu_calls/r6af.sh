#!/bin/bash
# round 6, call af: delta marks with the map still read as four 16-byte words per lane (call ae: the reads had been narrowed to eight
# ds_read2_b32 at a lane stride of 64 bytes): the chase alone | + delta marks, 100 registers | + delta marks, 96 registers (5 per SIMD)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in chase nowpe ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  for r in 1 2; do ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"; done
done > gpurun_out/r6af_exec_probe.txt 2>&1
cat gpurun_out/r6af_exec_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_exec_seg.py tests/test_gpu_decode.py tests/test_gpu_generated_frames.py -q -x --timeout 900 2>&1 | tail -4
