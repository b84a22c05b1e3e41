#!/bin/bash
# round 6, call aq: counters of the two entropy kernels of the headline archive (zk_k_huf, zk_k_fse_predef_fed) and the executor
rm -rf gpurun_out/pmcx gpurun_out/pmcx_*.log
KERNELS=huf,fse,exec bash tools/pmc_exec.sh 2048 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r6aq_pmc_entropy.txt
cat gpurun_out/r6aq_pmc_entropy.txt
rm -rf gpurun_out/pmcx gpurun_out/pmcx_*.log
