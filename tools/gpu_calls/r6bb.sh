#!/bin/bash
# round 6, call bb: counters of zk_k_fse_quad<ZkCellsX16, 56, 4> on the reference-made level-1 archive (4 GiB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcq
Q="--steps 1 --warmup 1 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --sync --choice xxh64=5 --cache /tmp/zkcache --archive libzstd --level 1"
python bench.py $Q > /dev/null 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcq/$tag -o p --output-format csv -- python bench.py $Q --no-fork > gpurun_out/pmcq_$tag.log 2>&1 || echo "pass failed: $set"
done
KERNELS=fse_quad,huf,exec python tools/pmc_table.py gpurun_out/pmcq/* > gpurun_out/r6bb_pmc_quad.txt
cat gpurun_out/r6bb_pmc_quad.txt
rm -rf gpurun_out/pmcq gpurun_out/pmcq_*.log
