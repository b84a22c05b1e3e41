#!/usr/bin/env python3
"""Scratch: per-phase shader-clock totals inside zk_k_enc_match (library built with -DZKE_CLOCKS: tools/build_enc_variant.sh clk:-DZKE_CLOCKS).
   ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_clk.so python tools/enc_clocks.py [frames] [level]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
F = 2 << 20
dev = torch.device("cuda:0")
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(64 * F), np.uint8)
d_src = torch.from_numpy(np.tile(data, nf // 64)).to(dev)
n = nf * F
cap = int(zk.lib.zk_compress_bound(n, F))
d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
d_cs = torch.zeros(nf, dtype=torch.int32, device=dev); d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
raw = C.CDLL(zk.LIB_PATH)
raw.zk_debug_enc_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
eng.set_profiling(True)
eng.encode_frames_dev(d_src, n, F, level, False, d_comp, cap, d_cs, d_ds)
raw.zk_debug_enc_clocks(None, 1)
_, csize = eng.encode_frames_dev(d_src, n, F, level, False, d_comp, cap, d_cs, d_ds)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
raw.zk_debug_enc_clocks(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["0 segment start (table, ring, history)", "1 ring words + hashes + far lookups", "2 wait barrier 1", "3 insert", "4 wait barrier 2",
         "5 comparisons -> best[]", "6 tile summary", "7 -", "8 near lookups + stitch + stores of the group before", "9 parse A: entries, masks, next table",
         "10 parse B: the walk", "11 parse C1: starts + sequences", "12 parse C2: literal bytes",
         "13 zk_k_enc_dense_cand: lists into the tables", "14 zk_k_enc_dense_cand: lookups", "15 zk_k_enc_dense_cand: last sweep"]
tot = v.sum()
groups = n / 4096
print("match ms", round(eng.kernel_times()["zk_k_enc_match"], 3), "ratio", round(n / csize, 3), "level", level)
for nm, x in zip(names, v):
    print(f"  {nm:40s} {x / tot * 100:5.1f} %   {x / groups / 16:9.0f} clocks per group and wave")
print(f"  {'total':40s}         {tot / groups / 16:9.0f}")
