#!/usr/bin/env python3
"""Scratch: encode kernel times on inputs other than the generator's text (zeros, short periods, mixed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = 2 << 20
dev = torch.device("cuda:0")
eng = zk.Engine(0)
rng = np.random.default_rng(7)
def mk(kind):
    if kind == "text": return np.frombuffer(zko.gen_chunks(64 * F), np.uint8)
    if kind == "zeros": return np.zeros(64 * F, np.uint8)
    if kind == "random": return rng.integers(0, 256, 64 * F, dtype=np.uint8)
    if kind == "period7": return np.tile(np.arange(7, dtype=np.uint8) * 31 + 1, 64 * F // 7 + 1)[:64 * F].copy()
    if kind == "period300": return np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 64 * F // 300 + 1)[:64 * F].copy()
    if kind == "mixed":
        t = np.frombuffer(zko.gen_chunks(64 * F), np.uint8).copy()
        r = rng.integers(0, 256, 64 * F, dtype=np.uint8)
        for i in range(0, 64 * F, 1 << 16):
            if (i >> 16) % 3 == 1: t[i:i + (1 << 16)] = r[i:i + (1 << 16)]
            if (i >> 16) % 3 == 2: t[i:i + (1 << 16)] = 0
        return t
    raise ValueError(kind)
for kind in sys.argv[2:] or ["text", "zeros", "random", "period7", "period300", "mixed"]:
    d_src = torch.from_numpy(np.tile(mk(kind), nf // 64)).to(dev)
    n = nf * F
    cap = int(zk.lib.zk_compress_bound(n, F))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
    eng.set_profiling(True)
    for r in range(2):
        _, csize = eng.encode_frames_dev(d_src, n, F, 1, True, d_comp, cap, d_cs, d_ds)
    print(f"{kind:10s} ratio {n / csize:9.2f}", {k.replace("zk_k_enc_", ""): round(v, 2) for k, v in eng.kernel_times().items()}, flush=True)
    del d_src, d_comp
