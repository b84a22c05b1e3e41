#!/bin/bash
# round 4, call h: what is a step of the quad walk made of?  knock-outs and the aligned reader, zk_k_fse on 4 GiB of libzstd frames
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in "" koload kodpp reva; do
  if [ -z "$v" ]; then unset ZEEKSTD_AMD_LIB; else export ZEEKSTD_AMD_LIB=$PWD/tools/variants/libzk_$v.so; fi
  echo "== ${v:-product}"; timeout 200 python tools/fse_probe.py 256 0 16 2>&1 | grep FSEPROBE
done > gpurun_out/r4h_fse_knockouts.log 2>&1
cat gpurun_out/r4h_fse_knockouts.log
