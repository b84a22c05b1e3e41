#!/bin/bash
# round 4, call m: kernel traces of the decode with the checksums beside the executor (what overlaps what?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/follow_probe.py --steps 2 --only default --cache /tmp/zkcache > /dev/null 2>&1      # fills the input cache (forked generation)
for v in default follow_resident4 follow; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04m_$v -- python tools/follow_probe.py --steps 3 --only $v --no-fork --cache /tmp/zkcache > gpurun_out/r04m_$v.json 2> gpurun_out/r04m_$v.err
  echo "== $v"; python tools/prof_timeline.py gpurun_out/prof_r04m_$v 5 > gpurun_out/r04m_${v}_timeline.txt; tail -42 gpurun_out/r04m_${v}_timeline.txt
  rm -rf gpurun_out/prof_r04m_$v
done
