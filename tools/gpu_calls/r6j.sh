#!/bin/bash
# round 6, call j: batches without checksums launch no checksum kernel; 8-byte cells for the sequence chains of small batches; configs[0] again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_exec_seg.py tests/test_gpu_kernel_choice.py tests/test_gpu_decode.py tests/test_gpu_decoder_api.py tests/test_gpu_generated_frames.py -x -q --timeout 600 2>&1 | tail -5
timeout 300 python tools/c0_probe.py 2>&1 | tail -8 | tee gpurun_out/r6j_c0_probe.txt
timeout 300 python tools/seg_probe.py --frames 1,16 --seg-kib 128 2>&1 | grep "frame \|seg128 " | tee gpurun_out/r6j_seg_probe_gpu.txt
