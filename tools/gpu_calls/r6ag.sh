#!/bin/bash
# round 6, call ag: the map's 16-byte chunks permuted inside every 128 bytes (chunk c at c ^ ((c >> 3) & 7)): a lane's four accesses of 16 bytes no longer
# hit two groups of banks eight lanes at a time: the library of call af (chase + delta marks) | this tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in af "" af ""; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  for r in 1; do ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 2>&1 | grep EXECVAR | sed "s|^|$v |"; done
done > gpurun_out/r6ag_exec_probe.txt 2>&1
cat gpurun_out/r6ag_exec_probe.txt
