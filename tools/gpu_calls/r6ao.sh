#!/bin/bash
# round 6, call ao: the bench line with the executor of call al and the fed checksum pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py 2>gpurun_out/r6ao_bench.err | tail -1 > gpurun_out/r6ao_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6ao_bench.json"))
keys = ["value", "ms_per_step", "reference_made_GiB_s", "reference_made_level3_GiB_s", "encode_GiB_s", "round_trip_GiB_s", "seek_p50_us", "seek_p50_us_reference_made", "configs0_decoder_GiB_s", "roofline_frac"]
print({k: d.get(k) for k in keys})
print(d.get("roofline"))
PY
tail -3 gpurun_out/r6ao_bench.err
