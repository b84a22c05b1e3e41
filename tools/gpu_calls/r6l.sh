#!/bin/bash
# round 6, call l: bench.py at 8 ranks on the one GPU: what the root finds in the gathered archive
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
df -h /dev/shm | tail -1
ZK_BENCH_DEBUG=1 HIP_VISIBLE_DEVICES=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --frames 64 --steps 2 --warmup 1 --no-cpu-baseline --no-seek --no-e2e --no-c1 --one-gpu-transport tests/sim/libzk_shm_collectives.so 2>&1 | grep -v Gloo | tail -22 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps(d['rccl_gather'])[:1500]); print('setup_s', d['setup_s'])
    else: print(l[:300])
"
