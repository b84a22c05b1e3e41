#!/bin/bash
# round 5, call j: what the driver runs at round end + the rocprofv3 evidence (kernel trace, FETCH / WRITE passes on three legs), the matcher's SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_round_end.sh r05 ref 2>&1 | tail -120
bash tools/pmc_enc.sh 1024 > gpurun_out/r05_pmc_enc_match_raw.txt 2>&1; tail -30 gpurun_out/r05_pmc_enc_match_raw.txt
