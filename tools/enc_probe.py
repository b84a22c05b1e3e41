#!/usr/bin/env python3
"""Scratch: encode-kernel timings (frames x 2 MiB of the generator text)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
for r in range(2):
    _, csize = eng.encode_frames_dev(d_src, n, F, 1, False, d_comp, cap, d_cs, d_ds)
print("MATCHVAR", os.environ.get("ZK_MATCH_VARIANT", "0"), "ratio", round(n / csize, 3), {k: round(v, 3) for k, v in eng.kernel_times().items()}, flush=True)
