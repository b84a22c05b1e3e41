#!/bin/bash
# round 6, call n: level-3 reference-made archive, two batches in flight: 256-lane tiles (ring of 4 T) against 512-lane tiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-seek --no-ref-archive --no-e2e --no-c1 --cache /tmp/zkcache --archive libzstd --level 3"
for ch in "" "--choice exec_lanes=512"; do
  python bench.py $Q $ch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$ch', d['value'], d['ms_per_step'], d['one_batch_at_a_time']['ms_per_step'], d['roofline']['kernel_ms'])"
done | tee gpurun_out/r6n_l3_lanes.txt
