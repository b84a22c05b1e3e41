#!/bin/bash
# round 6, call cl: open-ended campaigns on the round's last tree -- the decoder on libzstd-made archives of random structured inputs, damaged archives
# through the Level-C shim, the GPU encoder against its twin (levels 1 / 2 / 3 / 6 / 0 / 9), single seeks on a level-3 archive this engine wrote
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== fuzz_decode_gpu 1500 (seed 61)"; timeout 900 python tools/fuzz_decode_gpu.py 1500 61 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -3
echo "== fuzz_levelc_gpu 1500 (seed 62)"; timeout 900 python tools/fuzz_levelc_gpu.py 1500 62 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -3
echo "== fuzz_encode_gpu 6000 (seed 63)"; timeout 1200 python tools/fuzz_encode_gpu.py 6000 63 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -3
echo "== seek_soak 256 MiB, 20000 seeks, level 3, checksums, engine-made"; timeout 900 python tools/seek_soak.py 256 20000 3 1 1 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -4
} | tee gpurun_out/r06f_campaigns.txt
