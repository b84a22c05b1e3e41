#!/bin/bash
# round 6, call t: the small path with the executor in segments: the suites that read through the handles, seeks into 2 MiB-frame archives
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_seeks.py tests/test_gpu_decoder_api.py tests/test_gpu_host_pipeline.py tests/test_gpu_levelc.py tests/test_gpu_exec_seg.py -x -q --timeout 900 2>&1 | tail -4
timeout 600 python tools/seek2m_probe.py 128 300 2>&1 | tail -10 | tee gpurun_out/r6t_seek2m_probe.txt
timeout 300 python tools/c0_probe.py 2>&1 | tail -6 | tee gpurun_out/r6t_c0_probe.txt
