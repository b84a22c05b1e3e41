#!/usr/bin/env python3
"""Scratch: step time of the decode with one / two batches in flight, with and without checksum verification."""
import os, sys, time
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
outs = [torch.empty(n + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
sts = [torch.zeros(nf, dtype=torch.int32, device=dev) for _ in range(2)]
for verify in (True, False):
    for depth in (1, 2):
        def run(k):
            pend = []
            for i in range(k):
                if len(pend) == depth:
                    assert eng.decode_wait(pend.pop(0)) == 0
                pend.append(eng.decode_submit_dev(d_comp, csize, d_c, d_d, 0, nf, outs[i & 1], n, verify, sts[i & 1]))
            for s in pend:
                assert eng.decode_wait(s) == 0
        run(4)
        torch.cuda.synchronize(); t = time.perf_counter(); run(12); torch.cuda.synchronize()
        print(f"verify={verify} in_flight={depth}: {(time.perf_counter() - t) / 12 * 1e3:.2f} ms/step", flush=True)
