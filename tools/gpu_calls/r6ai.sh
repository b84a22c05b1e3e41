#!/bin/bash
# round 6, call ai: counters of the executor as it is now (global gathers, short chase, delta marks, swizzled map)
rm -rf gpurun_out/pmcx gpurun_out/pmcx_*.log
bash tools/pmc_exec.sh 2048 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r6ai_pmc_exec.txt
cat gpurun_out/r6ai_pmc_exec.txt
rm -rf gpurun_out/pmcx
