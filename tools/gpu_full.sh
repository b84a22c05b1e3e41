#!/bin/bash
# full -m gpu suite + the default bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > gpurun_out/r3_${tag}_gpu_tests.log; cat gpurun_out/r3_${tag}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r3_${tag}_bench.json 2> gpurun_out/r3_${tag}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3_${tag}_bench.err
python - <<PY
import json
l=json.load(open("gpurun_out/r3_${tag}_bench.json"))
print({k:l[k] for k in ("value","ms_per_step")}, l["encode"], l["round_trip"], l["roofline"]["frac"], l["reference_made_archive"]["value"], l["reference_made_archive"].get("level_3",{}).get("value"), l["seek"] and {k:v for k,v in l["seek"].items() if "p50" in str(k) or isinstance(v,(int,float))})
PY
