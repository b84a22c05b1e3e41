#!/usr/bin/env python3
"""Scratch: encode + decode rates on inputs other than the generator's text (cliffs: zeros, random bytes, short periods, long literal runs)."""
import os, sys, time
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
    if kind == "period3": return np.tile(np.array([200, 30, 77], np.uint8), 64 * F // 3 + 1)[:64 * F].copy()
    if kind == "period300": return np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 64 * F // 300 + 1)[:64 * F].copy()
    if kind == "nibbles": return rng.integers(0, 16, 64 * F, dtype=np.uint8)                       # literals only, 4 bits of entropy
    if kind == "mixed":
        t = np.frombuffer(zko.gen_chunks(64 * F), np.uint8).copy()
        r = rng.integers(0, 256, 64 * F, dtype=np.uint8)
        for i in range(0, 64 * F, 1 << 16):
            if (i >> 16) % 3 == 1: t[i:i + (1 << 16)] = r[i:i + (1 << 16)]
            if (i >> 16) % 3 == 2: t[i:i + (1 << 16)] = 0
        return t
    raise ValueError(kind)
for kind in sys.argv[2:] or ["text", "zeros", "random", "period7", "period300", "nibbles", "mixed"]:
    data = mk(kind)
    d_src = torch.from_numpy(np.tile(data, nf // 64)).to(dev)
    n = nf * F
    cap = int(zk.lib.zk_compress_bound(n, F))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
    eng.set_profiling(False)
    eng.encode_frames_dev(d_src, n, F, 1, True, d_comp, cap, d_cs, d_ds)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, csize = eng.encode_frames_dev(d_src, n, F, 1, True, d_comp, cap, d_cs, d_ds)
    torch.cuda.synchronize(); te = time.perf_counter() - t0
    cs = d_cs.cpu().numpy().astype(np.uint64)
    c = np.zeros(nf + 1, np.uint64); d = np.zeros(nf + 1, np.uint64); c[1:] = np.cumsum(cs); d[1:] = np.cumsum(np.full(nf, F, np.uint64))
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
    rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, True, d_st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, True, d_st)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    eng.set_profiling(True)
    eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, True, d_st)
    kt = {k.replace("zk_k_", ""): round(v, 2) for k, v in eng.kernel_times().items() if v >= 0.05}
    ok = bool(torch.equal(d_out[:n], d_src))
    print(f"{kind:10s} ratio {n / csize:8.2f}  encode {n / 2**30 / te:7.1f} GiB/s  decode {n / 2**30 / td:7.1f} GiB/s  rc {rc} ok {ok}  {kt}", flush=True)
    del d_src, d_comp, d_out
