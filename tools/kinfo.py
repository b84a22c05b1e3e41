#!/usr/bin/env python3
"""Scratch: registers / LDS / scratch of every kernel in a HIP source (compiles the device side to assembly).
   tools/kinfo.py zeekstd_amd/csrc/zk_decode.hip [-DNAME=V ...]"""
import re, subprocess, sys, tempfile, os
src, flags = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + flags,
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
meta = text[text.index("amdhsa.kernels:"):]
for ent in re.split(r"\n  - \.", meta)[1:]:
    f = dict(re.findall(r"\.?(\w+):\s+(\S+)", ent))
    name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"{name:44s} vgpr {f.get('vgpr_count'):>4} sgpr {f.get('sgpr_count'):>4} lds {f.get('group_segment_fixed_size'):>7} scratch {f.get('private_segment_fixed_size'):>5} spill {f.get('vgpr_spill_count')}")
