#!/usr/bin/env python3
"""Scratch GPU bring-up script: decode parity on assorted libzstd-made archives + a first timing.
Run on the GPU box:  python tools/gpu_check.py [--big MiB]"""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import zko, libzstd_ref as Z
import zeekstd_amd as zk

ap = argparse.ArgumentParser()
ap.add_argument("--big", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--noparity", action="store_true")
args = ap.parse_args()

eng = zk.Engine(0)
print("device:", eng.device_name, "libzstd system:", Z.version("system"), flush=True)

def offs(frames):
    c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
    return c, d

# xxh64
datas = [b"", b"a", b"Hello, World!", bytes(range(256)), zko.gen_text(100000, 7), zko.gen_text(2 << 20, 0x5EED0002)]
blob = b"".join(datas); o = np.zeros(len(datas) + 1, np.uint64); o[1:] = np.cumsum([len(x) for x in datas])
h = eng.xxh64_frames(blob, o)
exp = [zko.xxh64(x) for x in datas]
print("xxh64:", "OK" if list(map(int, h)) == exp else ("FAIL", [hex(int(x)) for x in h], [hex(x) for x in exp]), flush=True)

random.seed(1)
txt = zko.gen_chunks(5 << 20)
cases = [("text", txt), ("zeros", bytes(3 << 20)), ("rand", random.randbytes(1 << 20)),
         ("mixed", random.randbytes(100000) + txt[:300000] + bytes(50000) + b"abc" * 30000 + txt[100000:400000] + b"ab" * 5000 + b"x" * 70000)]
nfail = 0
for name, data in ([] if args.noparity else cases):
    for level in ((1, 3) if args.quick else (-5, 1, 3, 9, 19)):
        for fs in (2 << 20, 65536, 1000, 100):
            d = data[:50000] if (fs <= 1000 and len(data) > 200000) else data
            if level >= 9 and len(d) > (1 << 20): d = d[:1 << 20]
            for cks in (False, True):
                comp, frames = Z.encode_seekable_frames(d, fs, level, cks, "system")
                c, dd = offs(frames)
                try:
                    out, st = eng.decode_frames(comp + b"\0" * 8, c, dd, verify=True, raise_on_error=False)
                except Exception as e:
                    print("EXC", name, level, fs, cks, e, flush=True); nfail += 1; continue
                if out != d or st.any():
                    nfail += 1
                    a = np.frombuffer(out, np.uint8); b = np.frombuffer(d, np.uint8)
                    mm = np.nonzero(a != b)[0]
                    print("FAIL", name, level, fs, cks, "status", st[st != 0][:5], "nmis", mm.size, "first", mm[:3], flush=True)
    print("case", name, "done, fails so far", nfail, flush=True)
print("PARITY", "ALL OK" if nfail == 0 else f"{nfail} FAILURES", flush=True)

# timing
import torch
n = args.big << 20
t = time.time(); data = zko.gen_chunks(n); print("gen", round(time.time() - t, 2), "s", flush=True)
t = time.time(); comp, frames = Z.encode_seekable_frames(data, 2 << 20, 1, True, "system"); tenc = time.time() - t
print(f"cpu encode {n/tenc/2**20:.1f} MiB/s ratio {n/len(comp):.3f}", flush=True)
t = time.time(); ref = Z.decode_stream(comp, n, "system"); tdec = time.time() - t
print(f"cpu decode {n/tdec/2**30:.3f} GiB/s", flush=True)
c, dd = offs(frames)
dev = torch.device("cuda:0")
d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(dd.view(np.int64)).to(dev)
d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev)
d_st = torch.zeros(len(frames), dtype=torch.int32, device=dev)
for verify in (False, True):
    ts = []
    for r in range(args.reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, len(frames), d_out, n, verify, d_st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ok = bytes(d_out[:n].cpu().numpy()) == data
    print(f"gpu decode verify={verify} rc={rc} ok={ok} best {n/min(ts)/2**30:.2f} GiB/s  times(ms) {[round(x*1e3,2) for x in ts]}", flush=True)
