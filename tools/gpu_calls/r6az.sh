#!/bin/bash
# round 6, call az: tools/gpu_round_end.sh r06b ref -- the whole suite, smoke, the driver's bench line, kernel traces + FETCH / WRITE passes on the three legs
bash tools/gpu_round_end.sh r06b ref 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -150
