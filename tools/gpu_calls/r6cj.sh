#!/bin/bash
# round 6, call cj: the Huffman kernel's table pool, 8192 | 4096 | 2560 cells per 16 blocks (LDS per workgroup 25.8 | 17.6 | 14.5 KiB): the kernels alone
# (tools/exec_probe.py) and the pipelined step (bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --cache /tmp/zkcache"
python bench.py $Q > /dev/null 2>&1
for v in "" p4096 p2560 "" p4096 p2560; do
  lib=zeekstd_amd/libzk_$v.so; [ -z "$v" ] && lib=zeekstd_amd/libzeekstd_amd.so
  ZEEKSTD_AMD_LIB=$PWD/$lib python tools/exec_probe.py 2048 verify 2>&1 | grep EXECVAR | sed "s|^|pool ${v:-8192} |"
  ZEEKSTD_AMD_LIB=$PWD/$lib python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pool ${v:-8192} bench', d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r6cj_huf_pool.txt
