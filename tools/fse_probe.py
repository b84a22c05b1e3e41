#!/usr/bin/env python3
"""Scratch: sequence-kernel timings on an archive written by the box's libzstd (own-table blocks).
usage: fse_probe.py <MiB> <mode 0|1|2>   (mode: zk_engine_set_fse_kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
from oracle import libzstd_ref as Z
import zeekstd_amd as zk
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
F = 2 << 20
dev = torch.device("cuda:0")
eng = zk.Engine(0)
data = zko.gen_chunks(mib << 20)
comp, frames = Z.encode_seekable_frames(data, F, 1, True, "system")
comp = comp * rep; frames = frames * rep
nf = len(frames)
c = np.zeros(nf + 1, np.uint64); d = np.zeros(nf + 1, np.uint64)
c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
n = int(d[-1])
d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
eng.set_profiling(True); eng.set_fse_kernel(mode)
res = []
for r in range(3):
    eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, True, d_st)
    k = eng.kernel_times(); res.append(round(k['zk_k_fse'], 3))
ok = bytes(d_out[:len(data)].cpu().numpy()) == data and not d_st.cpu().numpy().any()
print("FSEPROBE MiB", mib * rep, "frames", nf, "mode", mode, "fse_ms", res, "bit_exact", ok, flush=True)
