#!/usr/bin/env python3
"""Scratch: decode-kernel timings on a GPU-made archive (frames x 2 MiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
verify = len(sys.argv) > 2 and sys.argv[2] == "verify"
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
eng.set_profiling(True)
if os.environ.get("ZK_LANES"): eng.set_kernel_choice(exec_lanes=int(os.environ["ZK_LANES"]))
if os.environ.get("ZK_XXH"): eng.set_kernel_choice(xxh64=int(os.environ["ZK_XXH"]))
for r in range(3):
    rc = eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, n, verify, d_st)
print("EXECVAR", "lanes=" + os.environ.get("ZK_LANES", "auto"), "xxh=" + os.environ.get("ZK_XXH", "auto"), "rc", rc, "ok", bool(torch.equal(d_out[:n], d_src)) and int(d_st.abs().sum()) == 0, {k: round(v, 3) for k, v in eng.kernel_times().items()}, flush=True)
