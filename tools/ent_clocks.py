#!/usr/bin/env python3
"""Scratch: per-phase, per-wave shader-clock totals inside zk_k_enc_entropy, the shader clock and how many workgroups share a CU
   (tools/build_enc_variant.sh eclk:-DZKE_ENT_CLOCKS).
   ZEEKSTD_AMD_LIB=zeekstd_amd/libzk_eclk.so python tools/ent_clocks.py [frames] [level]"""
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
eng.encode_frames_dev(d_src, n, F, level, False, d_comp, cap, d_cs, d_ds)
torch.cuda.synchronize()
out = (C.c_ulonglong * 48)()
raw.zk_debug_enc_clocks(out, 0)
res = list(out)[40:48]
v = np.array(list(out)[:40], dtype=np.float64).reshape(8, 5)
names = ["clear + tables + raw / literal counts", "wait for the helper's counts", "-", "huffman builds", "bit writers (wave 0 literals, 1 FSE chains, 2 emitter)", "wait at the barrier", "layout + piece tables", "-"]
wgs = n / (16 * 32768)
print("entropy ms", round(eng.kernel_times()["zk_k_enc_entropy"], 3))
print(f"  {'phase':44s}" + "".join(f"   wave {w}" for w in range(5)) + "   (clocks per workgroup)")
for nm, row in zip(names, v):
    if nm != "-":
        print(f"  {nm:44s}" + "".join(f" {x / wgs:8.0f}" for x in row))
print(f"  shader clock during the kernel: {v[7][0] / max(v[7][1], 1) * 100:.0f} MHz (clock64 / wall_clock64 of wave 0, 100 MHz reference)")
print(f"  {'total':44s}" + "".join(f" {x / wgs:8.0f}" for x in v[:7].sum(axis=0)))
print("  workgroups already on the CU when one starts (0, 1, 2, ...):", [int(x) for x in res])
