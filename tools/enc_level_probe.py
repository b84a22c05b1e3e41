#!/usr/bin/env python3
"""Scratch: encode-kernel timings per level (frames x 2 MiB of the generator text): python tools/enc_level_probe.py [frames] [levels...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
levels = [int(x) for x in sys.argv[2:]] or [1, 3, 6]
F = 2 << 20
dev = torch.device("cuda:0")
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(64 * F), np.uint8)
d_src = torch.from_numpy(np.tile(data, nf // 64)).to(dev)
n = nf * F
cap = int(zk.lib.zk_compress_bound(n, F))
d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
eng.set_profiling(True)
for level in levels:
    for r in range(2):
        _, csize = eng.encode_frames_dev(d_src, n, F, level, False, d_comp, cap, d_cs, d_ds)
    print("LEVEL", level, "ratio", round(n / csize, 3), {k: round(v, 3) for k, v in eng.kernel_times().items()}, flush=True)
