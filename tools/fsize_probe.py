#!/usr/bin/env python3
"""Scratch: encode + decode rates of 1 GiB of generator text against the frame size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
total = 1 << 30
dev = torch.device("cuda:0")
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(128 << 20), np.uint8)
d_src = torch.from_numpy(np.tile(data, total // len(data))).to(dev)
for fs in [int(x) for x in sys.argv[1:]] or [4096, 65536, 1 << 20, 2 << 20, 16 << 20, 128 << 20]:
    nf = total // fs
    cap = int(zk.lib.zk_compress_bound(total, fs))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
    eng.set_profiling(False)
    eng.encode_frames_dev(d_src, total, fs, 1, True, d_comp, cap, d_cs, d_ds)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, csize = eng.encode_frames_dev(d_src, total, fs, 1, True, d_comp, cap, d_cs, d_ds)
    torch.cuda.synchronize(); te = time.perf_counter() - t0
    cs = d_cs.cpu().numpy().astype(np.uint64)
    c = np.zeros(nf + 1, np.uint64); d = np.zeros(nf + 1, np.uint64); c[1:] = np.cumsum(cs); d[1:] = np.cumsum(np.full(nf, fs, np.uint64))
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.empty(total + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
    rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, total, True, d_st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, total, True, d_st)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    eng.set_profiling(True)
    eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, total, True, d_st)
    kt = {k.replace("zk_k_", ""): round(v, 2) for k, v in eng.kernel_times().items() if v >= 0.05}
    ok = bool(torch.equal(d_out[:total], d_src))
    print(f"frame {fs:>10d} x {nf:>6d}  ratio {total / csize:5.2f}  encode {1 / te:6.1f} GiB/s  decode {1 / td:6.1f} GiB/s  rc {rc} ok {ok}  {kt}", flush=True)
    del d_comp, d_out
