#!/bin/bash
# round 6, call aa: a wave per frame behind the executor's progress words (zk_k_xxh64_follow1) for a handful of verified frames
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_exec_seg.py tests/test_gpu_decode.py tests/test_gpu_decoder_api.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
timeout 300 python tools/seg_probe.py --frames 1,5,16,32 --seg-kib 128 2>&1 | grep "verify 1" | grep "frame \|seg128 " | tee gpurun_out/r6aa_follow1_probe.txt
