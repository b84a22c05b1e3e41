#!/bin/bash
# PMC passes over tools/enc_probe.py (each pass = its own run; counters only, no trace domains beside kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1024}
rm -rf gpurun_out/pmce
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmce/$tag -o p --output-format csv -- python tools/enc_probe.py $N > gpurun_out/pmce_$tag.log 2>&1 || echo "pass failed: $set"
done
KERNELS=enc_match python tools/pmc_table.py gpurun_out/pmce/*
