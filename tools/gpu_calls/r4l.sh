#!/bin/bash
# round 4, call l: the checksums beside the executor (zk_k_xxh64_follow) and the executor at 93 registers -- parity of the variants, then timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernel_choice.py -k "follow or beside or unknown" -x -q --timeout 600 --durations=5 2>&1 | tail -14 > gpurun_out/r04l_tests.log; cat gpurun_out/r04l_tests.log
timeout 600 python tools/follow_probe.py --steps 6 > gpurun_out/r04l_follow_probe.json 2> gpurun_out/r04l_follow_probe.err; echo "probe rc=$?"; cat gpurun_out/r04l_follow_probe.err | tail -12
