#!/bin/bash
# round 6, call v: zk_k_exec_fill_lds with a segment's turn as two trips to memory (descriptors read ahead, the image by LDS DMA)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_exec_seg.py tests/test_gpu_decoder_api.py tests/test_gpu_seeks.py -x -q --timeout 900 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
timeout 300 python tools/seg_probe.py --frames 1,16 --seg-kib 128 2>&1 | grep "verify 0" | grep "frame \|seg128 " | tee gpurun_out/r6v_seg_probe.txt
timeout 300 python tools/seg_probe.py --frames 1 --seg-kib 128 --archive libzstd --level 3 2>&1 | grep "verify 0" | grep "frame \|seg128 " | tee -a gpurun_out/r6v_seg_probe.txt
timeout 300 python tools/c0_probe.py 2>&1 | grep "by shape" | tee -a gpurun_out/r6v_seg_probe.txt
