#!/bin/bash
# round 5, call x: the small path's upload without scratch memory (eight named values): seek tests + soak + the seek leg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seeks.py tests/test_gpu_seek_soak.py tests/test_gpu_decoder_api.py tests/test_gpu_host_pipeline.py -q --timeout 600 2>&1 | tail -3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_seek -- python tools/seek_probe.py 64 600 > gpurun_out/r5x_seek.log 2>&1
grep "made" gpurun_out/r5x_seek.log | cut -c1-200
python tools/prof_summary.py gpurun_out/prof_seek 8 | cut -c1-120 | sed -n 3,8p; python tools/prof_summary.py gpurun_out/prof_seek 8 | tail -8 | cut -c1-120
rm -rf gpurun_out/prof_seek
