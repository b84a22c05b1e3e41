#!/bin/bash
# round 6, call cm: zk_k_enc_dense_cand with four | eight list entries in flight per lane (kernel trace of a level-3 encode, twice each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sed 's/for lvl in (1, 2, 3, 6, 9):/for lvl in (3,):/' tools/level_probe.py > tools/_l3_probe.py
for v in e8 e16 e8 e16; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l3 -- python tools/_l3_probe.py > gpurun_out/r6cm_l3.txt 2>&1
  echo "== ${v:-base}"; python tools/prof_summary.py gpurun_out/prof_l3 2 | grep "zk_k_enc_dense\|zk_k_enc_match"; grep "ratio" gpurun_out/r6cm_l3.txt | cut -c1-60
  rm -rf gpurun_out/prof_l3
done | tee gpurun_out/r6cm_dense_e8.txt
rm -f tools/_l3_probe.py
