#!/usr/bin/env python3
"""Scratch: time the decode kernels (HIP events) on a libzstd-made archive; no parity checks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko, libzstd_ref as Z
import zeekstd_amd as zk
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 256 << 20
data = zko.gen_chunks(n)
comp, frames = Z.encode_seekable_frames(data, 2 << 20, 1, False, "system")
c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
dev = torch.device("cuda:0")
eng = zk.Engine(0)
d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(len(frames), dtype=torch.int32, device=dev)
eng.set_profiling(True)
for r in range(3):
    rc = eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, len(frames), d_out, n, False, d_st)
kt = eng.kernel_times()
ok = bytes(d_out[:n].cpu().numpy()) == data
print("VARIANT", os.environ.get("ZK_FSE_VARIANT", "0"), "rc", rc, "ok", ok, {k: round(v, 3) for k, v in kt.items()}, flush=True)
