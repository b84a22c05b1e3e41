#!/bin/bash
# round 4, call o: the checksums beside / behind the executor by batch size (frames of 2 MiB), then the tests that touch the policy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 16 64 128 512 1024 2048; do
  timeout 300 python tools/follow_probe.py --frames $n --steps 10 > gpurun_out/r04o_follow_probe_$n.json 2> gpurun_out/r04o_$n.err; echo "== $n frames rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r04o_follow_probe_$n.json"))
for k,v in d["choices"].items(): print(f"   {k:14s} one at a time {v['one_at_a_time_ms']:7.3f} ms   two in flight {v['two_in_flight_ms']:7.3f} ms   followed {v['followed_one_at_a_time'][-1]:5d} / {v['followed_two_in_flight'][-1]:5d}   parity {v['parity']}")
PY
done > gpurun_out/r04_follow_by_batch_size.txt 2>&1
cat gpurun_out/r04_follow_by_batch_size.txt
timeout 900 python -m pytest tests/test_gpu_kernel_choice.py tests/test_gpu_decoder_api.py tests/test_gpu_seeks.py tests/test_gpu_host_pipeline.py -x -q --timeout 600 2>&1 | tail -4
python - <<'PY'
import bench, zeekstd_amd as zk
eng = zk.Engine(0)
print("configs0", bench.small_input_leg(eng, zk))
PY
