#!/usr/bin/env python3
"""Scratch: where a single-frame decode (one seek) spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
eng = zk.Engine(0)
F = 65536
data = np.frombuffer(zko.gen_chunks(256 * F), np.uint8)
comp, frames = eng.encode_frames(data, F, 1, True)
c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
buf = np.frombuffer(comp + b"\0" * 8, np.uint8)
def one(i):
    lo, hi = int(c[i]), int(c[i + 1])
    cc = np.array([0, hi - lo], np.uint64); dd = np.array([0, F], np.uint64)
    return eng.decode_frames(buf[lo:hi + 8], cc, dd, verify=True)
for i in range(5): one(i)
ts = []
for i in range(200):
    t = time.perf_counter(); out, st = one(i % 256); ts.append((time.perf_counter() - t) * 1e6)
assert out == data[(199 % 256) * F:(199 % 256 + 1) * F].tobytes()
print("host decode_frames 1 frame: p50 %.1f us  p10 %.1f" % (np.percentile(ts, 50), np.percentile(ts, 10)))
eng.set_profiling(True)
acc = {}
for i in range(20):
    one(i)
    for k, v in eng.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 20
print({k: round(v * 1000, 1) for k, v in acc.items()}, "us; sum", round(sum(acc.values()) * 1000, 1))
eng.set_profiling(False)
# device-resident single frame
dev = torch.device("cuda:0")
d_comp = torch.from_numpy(buf.copy()).to(dev); d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
d_out = torch.empty(F + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
ts = []
for i in range(200):
    t = time.perf_counter(); eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, i % 256, 1, d_out, F, True, d_st); ts.append((time.perf_counter() - t) * 1e6)
print("dev decode 1 frame: p50 %.1f us" % np.percentile(ts, 50))
