#!/usr/bin/env python3
"""Scratch: per-phase shader-clock totals inside zk_k_exec (library built with -DZK_EXEC_CLOCKS, tools/build_variants.sh clk:-DZK_EXEC_CLOCKS).
   ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/exec_clocks.py [frames]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
F = 2 << 20
dev = torch.device("cuda:0")
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(64 * F), np.uint8)
d_src = torch.from_numpy(np.tile(data, nf // 64)).to(dev)
n = nf * F
cap = int(zk.lib.zk_compress_bound(n, F))
d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
_, csize = eng.encode_frames_dev(d_src, n, F, 1, True, d_comp, cap, d_cs, d_ds)
cs = d_cs.cpu().numpy().astype(np.uint64)
c = np.zeros(nf + 1, np.uint64); d = np.zeros(nf + 1, np.uint64); c[1:] = np.cumsum(cs); d[1:] = np.cumsum(np.full(nf, F, np.uint64))
d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
raw = C.CDLL(zk.LIB_PATH)
raw.zk_debug_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
eng.set_profiling(True)
eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, False, d_st)
raw.zk_debug_clocks(None, 1)
eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, False, d_st)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
raw.zk_debug_clocks(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["block setup + first staging", "mark", "barrier waits", "prefetch + long + slot words + map", "chase", "pack + store", "settle", "gather addresses + loads + their wait"]
tot = v.sum()
print("exec ms", round(eng.kernel_times()["zk_k_exec"], 3), "waves", nf * 4)
for nm, x in zip(names, v):
    if x: print(f"  {nm:38s} {x / tot * 100:5.1f} %   {x / (nf * 4) / 2048:9.0f} clocks per 1 KiB of a wave")
