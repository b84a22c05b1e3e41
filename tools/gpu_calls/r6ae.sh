#!/bin/bash
# round 6, call ae: what made call ad's executor slower -- the chase alone | chase + delta marks without the occupancy attribute (100 registers, 4 per SIMD) | with it (96 + spill)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in chase nowpe ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  for r in 1 2; do ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"; done
done > gpurun_out/r6ae_exec_probe.txt 2>&1
cat gpurun_out/r6ae_exec_probe.txt
